#!/bin/bash
# Round 6: the proof by kernel at BASELINE config 5's size (2^22) and at 2^24 — kernel trace + stats, the two HBM-traffic passes and the wait /
# occupancy counters (one PMC group per run, --kernel-trace only) — for the sources in the tree. usage: collect_r6_big.sh TAG [SPARTAN_OPTIONS]
set -u
export TMPDIR=/tmp
export BENCH_NO_GATHER_PROBE=1
TAG=${1:-base}; OPTS=${2:-}
[ -n "$OPTS" ] && export SPARTAN_OPTIONS="$OPTS"
R=$(pwd); O=$R/gpurun_out/r6big_$TAG; mkdir -p $O
largest() { find "$1" -name "$2" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
for s in 22 24; do
  steps=2; [ $s = 24 ] && steps=1
  B="python $R/bench.py --log2-cons $s --no-cpu-baseline --concurrent 0 --steps $steps --warmup 1 --no-side-metrics --no-strong"
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $O/stats$s -- $B > $O/stats$s.log 2>&1
  cd $R
  python profiles/summarize.py "$(largest $O/stats$s '*_results.db')" --detail k_msm_rows,k_msm_flat,k_msm_q,k_msm_reduce,k_ipa_round > $O/kernel_stats_2p$s.txt 2>$O/summarize$s.err
  rm -rf $O/stats$s
  [ $s = 24 ] && break
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- $B > $O/pmc_$c.log 2>&1
    cp "$(largest $O/pmc_$c '*counter_collection.csv')" $O/pmc_$c.csv; rm -rf $O/pmc_$c
  done
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_wait -- $B > $O/pmc_wait.log 2>&1
  cp "$(largest $O/pmc_wait '*counter_collection.csv')" $O/pmc_wait.csv; rm -rf $O/pmc_wait
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_tcc -- $B > $O/pmc_tcc.log 2>&1
  cp "$(largest $O/pmc_tcc '*counter_collection.csv')" $O/pmc_tcc.csv; rm -rf $O/pmc_tcc
  cd $R
  cp profiles/pmc_traffic.json /tmp/pmc_traffic_keep.json
  python profiles/pmc_summarize.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv r6_pmc_hbm_traffic_2p22_$TAG.txt > $O/pmc_hbm_traffic_2p22.txt 2>&1
  cp profiles/pmc_traffic.json $O/pmc_traffic_2p22.json
  cp /tmp/pmc_traffic_keep.json profiles/pmc_traffic.json
  python profiles/pmc_counters.py $O/pmc_wait.csv $O/pmc_tcc.csv $O/pmc_wait.csv > $O/pmc_kernels_2p22.txt 2>&1
  rm -f $O/pmc_*.csv
done
ls -la $O
