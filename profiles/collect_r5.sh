#!/bin/bash
# Round 5, final profile set for the committed sources (run from the repo root through gpurun): kernel trace + stats, the two HBM-traffic
# counter passes, occupancy / cache / wait counter passes, the default bench line (with the live oracle), per-entry-point wall time, the bench
# lines at the other sizes, and the same trace + FETCH/WRITE passes for the two alternative row-MSM forms (LDS-staged, ring) on the standalone
# launch shapes. Counter passes use --kernel-trace only, one group per run. Library options travel through SPARTAN_OPTIONS.
set -u
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r5prof
mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --concurrent 0 --steps 4 --warmup 1 --no-side-metrics --no-strong"
export BENCH_NO_GATHER_PROBE=1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- $B > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_tcc -- $B > $O/pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_wait -- $B > $O/pmc_wait.log 2>&1
# the three row-MSM forms on the standalone launch shapes of a 2^20 proof (bench/msm_lds_probe.py): trace + HBM traffic + wait counters
P="python $R/bench/msm_lds_probe.py 20"
rocprofv3 --kernel-trace --stats -d $O/forms_stats -- $P > $O/forms_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/forms_pmc_$c -- $P > $O/forms_pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/forms_pmc_wait -- $P > $O/forms_pmc_wait.log 2>&1
cd $R
unset BENCH_NO_GATHER_PROBE
largest() { find "$1" -name "$2" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
for d in $O/pmc_* $O/forms_pmc_*; do [ -d $d ] && cp "$(largest $d '*counter_collection.csv')" $d.csv; done
python profiles/summarize.py "$(largest $O/stats '*_results.db')" --detail k_ipa_round,k_msm_reduce,k_msm_rows,k_msm_flat,k_cubic_bind2_eval,k_cubic_bind_eval_batched_eq,k_cubic_eval_batched_eq > $O/kernel_stats.txt 2>$O/summarize.err
python profiles/summarize.py "$(largest $O/forms_stats '*_results.db')" --detail k_msm_rows,k_msm_flat,k_msm_lds,k_msm_ring > $O/forms_kernel_stats.txt 2>>$O/summarize.err
rm -rf $O/stats $O/forms_stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq $O/pmc_tcc $O/pmc_wait $O/forms_pmc_FETCH_SIZE $O/forms_pmc_WRITE_SIZE $O/forms_pmc_wait
cp profiles/pmc_traffic.json $O/pmc_traffic.before.json
python profiles/pmc_summarize.py $O/forms_pmc_FETCH_SIZE.csv $O/forms_pmc_WRITE_SIZE.csv r5_pmc_hbm_traffic_forms.txt > $O/pmc_hbm_traffic_forms.txt 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic_forms.json
python profiles/pmc_summarize.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv r5_pmc_hbm_traffic.txt > $O/pmc_hbm_traffic.txt 2>&1
python profiles/pmc_counters.py $O/pmc_sq.csv $O/pmc_tcc.csv $O/pmc_wait.csv > $O/pmc_kernels.txt 2>&1
python profiles/pmc_counters.py $O/forms_pmc_wait.csv $O/forms_pmc_wait.csv $O/forms_pmc_wait.csv > $O/forms_pmc_kernels.txt 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python bench.py > $O/bench_line.json 2> $O/bench_line.err
SPARTAN_OPTIONS=testing.unlock=1,host.callstats=1 timeout 300 $B > /dev/null 2> $O/callstats.err; grep callstats $O/callstats.err | tail -52 > $O/callstats.txt
for s in 16 18; do python bench.py --log2-cons $s --cpu-log2-cons 0 --no-cpu-baseline > $O/bench_line_2p$s.json 2> $O/bench_line_2p$s.err; done
python bench.py --log2-cons 22 --no-cpu-baseline --steps 8 > $O/bench_line_2p22.json 2> $O/bench_line_2p22.err
python bench.py --log2-cons 24 --no-cpu-baseline --steps 3 --concurrent 0 --no-side-metrics > $O/bench_line_2p24.json 2> $O/bench_line_2p24.err
# the small-memory configuration (LDS-staged row MSM + 10-bit tables for the latency kernels: 15 GB instead of 118 GB at 2^20)
SPARTAN_OPTIONS=msm.lds_bits=10,msm.form=1,msm.wbits=10 python bench.py --no-cpu-baseline --concurrent 0 --no-side-metrics > $O/bench_line_small_memory.json 2> $O/bench_line_small_memory.err
./bench/ubench_fpmul > $O/ubench_fpmul.txt 2>&1
ls -la $O
# A/B of the row-MSM forms in the proof at 2^22 (BASELINE config 5's size)
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=600 bash scripts/gpu_ab.sh r5prof_ab22 1 "wide:" "ring6:msm.form=2,bg.eighths=6" "lds6:msm.lds_bits=10,msm.form=1,bg.eighths=6" 2>&1 | tail -8 > $O/ab_2p22.txt
