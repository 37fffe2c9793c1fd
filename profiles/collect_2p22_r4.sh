#!/bin/bash
# Round 4: a 2^22 proof (BASELINE config 5's size) by kernel — durations (kernel trace + stats) and HBM traffic (two PMC passes, kernel trace only),
# so that the streaming kernels' TB/s can be read at the size where they have left the launch-sized regime.
export TMPDIR=/tmp
export BENCH_NO_GATHER_PROBE=1
R=$(pwd); O=$R/gpurun_out/p22r4; mkdir -p $O
B="python $R/bench.py --log2-cons 22 --no-cpu-baseline --concurrent 0 --steps 2 --warmup 1 --no-side-metrics --no-strong"
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- $B > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc_$c -name "*counter_collection.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  cp $f $O/pmc_$c.csv; rm -rf $O/pmc_$c
done
cd $R
db=$(find $O/stats -name "*_results.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
python profiles/summarize.py "$db" > $O/kernel_stats_2p22.txt 2>$O/summarize.err
rm -rf $O/stats
cp profiles/pmc_traffic.json /tmp/pmc_traffic_keep.json
python profiles/pmc_summarize.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv r4_pmc_hbm_traffic_2p22.txt > $O/pmc_hbm_traffic_2p22.txt 2>&1
cp /tmp/pmc_traffic_keep.json profiles/pmc_traffic.json   # the committed file describes the 2^20 headline, not this run
rm -f $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv
ls -la $O
