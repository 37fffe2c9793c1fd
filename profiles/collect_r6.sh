#!/bin/bash
# Round 6, the profile set of the sources in the tree (run from the repo root through gpurun):
#   2^20 (the headline): kernel trace + stats, the two HBM-traffic passes (-> pmc_traffic.json entry of the default configuration), wait / VALU /
#   L2 counter passes, the F_q bandwidth table, the default bench line (live oracle), per-entry-point wall time;
#   2^22 (BASELINE config 5): kernel trace, the two traffic passes (-> the 2^22 entry), wait counters, the F_q table, the bench line;
#   2^24: kernel trace + bench line;  the small-memory configuration at 2^20: traffic passes + bench line;
#   the queue form standalone (bench/msm_queue_probe.py) under the wait / VALU counters; the ALU-ceiling micro-benchmark.
# Counter passes use --kernel-trace only, one group per run. Library options travel through SPARTAN_OPTIONS.
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r6prof; mkdir -p $O
largest() { find "$1" -name "$2" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
export BENCH_NO_GATHER_PROBE=1
pmc() { # pmc NAME "COUNTERS" -- cmd...
  local name=$1 ctrs=$2; shift 3
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1 )
  cp "$(largest $O/$name '*counter_collection.csv')" $O/$name.csv 2>/dev/null; rm -rf $O/$name
}
trace() { # trace NAME -- cmd...  -> $O/NAME.db
  local name=$1; shift 2
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/$name -- "$@" > $O/$name.log 2>&1 )
  cp "$(largest $O/$name '*_results.db')" $O/$name.db 2>/dev/null; rm -rf $O/$name
}
DET=k_ipa_round,k_msm_reduce,k_msm_q,k_msm_flat,k_msm_rows,k_cubic_bind2_eval,k_cubic_bind_eval_batched_eq,k_cubic_eval_batched_eq,k_sc_bind_eval,k_sc_eval
for s in 20 22; do
  steps=4; [ $s = 22 ] && steps=2
  B="python $R/bench.py --log2-cons $s --no-cpu-baseline --concurrent 0 --steps $steps --warmup 1 --no-side-metrics --no-strong"
  trace stats$s -- $B
  python profiles/summarize.py $O/stats$s.db --detail $DET > $O/kernel_stats_2p$s.txt 2>$O/summarize$s.err
  pmc fetch$s FETCH_SIZE -- $B
  pmc write$s WRITE_SIZE -- $B
  pmc wait$s "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" -- $B
  [ $s = 20 ] && pmc tcc$s "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" -- $B
  python profiles/pmc_summarize.py $O/fetch$s.csv $O/write$s.csv r6_pmc_hbm_traffic_2p$s.txt $s "" > $O/pmc_hbm_traffic_2p$s.txt 2>&1
  python profiles/pmc_counters.py $O/wait$s.csv $([ $s = 20 ] && echo $O/tcc$s.csv) > $O/pmc_kernels_2p$s.txt 2>&1
  python profiles/fq_bandwidth.py $O/stats$s.db $O/fetch$s.csv $O/write$s.csv profiles/r6_kernel_resources.txt $O/wait$s.csv > $O/fq_bandwidth_2p$s.txt 2>&1
done
# the small-memory configuration (LDS-staged row MSM + 10-bit tables for the latency kernels) at 2^20: its own traffic entry
SM=msm.form=1,msm.lds_bits=10,msm.wbits=10
B20="python $R/bench.py --no-cpu-baseline --concurrent 0 --steps 4 --warmup 1 --no-side-metrics --no-strong"
SPARTAN_OPTIONS=$SM pmc fetch_sm FETCH_SIZE -- $B20
SPARTAN_OPTIONS=$SM pmc write_sm WRITE_SIZE -- $B20
python profiles/pmc_summarize.py $O/fetch_sm.csv $O/write_sm.csv r6_pmc_hbm_traffic_small_memory.txt 20 "$SM" > $O/pmc_hbm_traffic_small_memory.txt 2>&1
# 2^24: kernel trace only
trace stats24 -- python $R/bench.py --log2-cons 24 --no-cpu-baseline --concurrent 0 --steps 1 --warmup 1 --no-side-metrics --no-strong
python profiles/summarize.py $O/stats24.db --detail k_msm_q,k_msm_flat,k_msm_reduce > $O/kernel_stats_2p24.txt 2>>$O/summarize24.err
# the queue form standalone under the wait / VALU counters (and the forms it replaced beside it)
pmc probe_wait "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" -- python $R/bench/msm_queue_probe.py 22 12/2/32,8/2/32 h
python profiles/pmc_counters.py $O/probe_wait.csv --kernels k_msm_q,k_msm_rows,k_msm_flat > $O/probe_pmc_kernels.txt 2>&1
timeout 600 python bench/msm_queue_probe.py 20 12/2/32,8/2/32 w,h,d > $O/queue_probe_2p20.txt 2>&1
timeout 900 python bench/msm_queue_probe.py 22 12/2/32,8/2/32 w,h > $O/queue_probe_2p22.txt 2>&1
unset BENCH_NO_GATHER_PROBE
cp profiles/pmc_traffic.json $O/pmc_traffic.json
# bench lines (they read the traffic entries written above)
python bench.py > $O/bench_line.json 2> $O/bench_line.err
SPARTAN_OPTIONS=testing.unlock=1,host.callstats=1 timeout 300 python bench.py --no-cpu-baseline --concurrent 0 --steps 4 --warmup 1 --no-side-metrics --no-strong > /dev/null 2> $O/callstats.err; grep callstats $O/callstats.err | tail -56 > $O/callstats.txt
for s in 16 18; do python bench.py --log2-cons $s --cpu-log2-cons 0 --no-cpu-baseline > $O/bench_line_2p$s.json 2> $O/bench_line_2p$s.err; done
python bench.py --log2-cons 22 --no-cpu-baseline --steps 8 > $O/bench_line_2p22.json 2> $O/bench_line_2p22.err
python bench.py --log2-cons 24 --no-cpu-baseline --steps 3 --concurrent 0 --no-side-metrics --no-strong > $O/bench_line_2p24.json 2> $O/bench_line_2p24.err
SPARTAN_OPTIONS=$SM python bench.py --no-cpu-baseline --concurrent 0 > $O/bench_line_small_memory.json 2> $O/bench_line_small_memory.err
./bench/ubench_fpmul > $O/ubench_fpmul.txt 2>&1
rm -f $O/*.db $O/fetch*.csv $O/write*.csv $O/wait*.csv $O/tcc*.csv $O/probe_wait.csv
ls -la $O
