#!/bin/bash
# Collects the round-3 profile set on a GPU box (run from the repo root through gpurun): kernel trace + stats, the two
# HBM-traffic counter passes, the occupancy / cache counter passes for the F_q kernels, and the default bench line.
# Counter passes use --kernel-trace only (no other trace domains), one counter group per run.
set -u
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r3prof
mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --concurrent 0 --steps 4 --warmup 1 --no-side-metrics --no-strong"
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- $B > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_tcc -- $B > $O/pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_wait -- $B > $O/pmc_wait.log 2>&1
cd $R
python bench.py > $O/bench_line.json 2> $O/bench_line.err
# keep what travels back small: the counter CSVs and the stats database summary
# (bench.py also runs bench/ubench_fpmul as a child process: every pass leaves one small file for it — take the largest)
largest() { find "$1" -name "$2" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
for d in $O/pmc_*; do [ -d $d ] && cp "$(largest $d '*counter_collection.csv')" $d.csv; done
python profiles/summarize.py "$(largest $O/stats '*_results.db')" > $O/kernel_stats.txt 2>$O/summarize.err
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq $O/pmc_tcc $O/pmc_wait
ls -la $O
