#!/bin/bash
# copies the collection of profiles/collect_r6.sh + collect_r6_diag.sh (gpurun_out/r6prof, scratch) to the committed names
O=gpurun_out/r6prof
for f in kernel_stats_2p20 kernel_stats_2p22 kernel_stats_2p24 pmc_hbm_traffic_2p20 pmc_hbm_traffic_2p22 pmc_hbm_traffic_small_memory pmc_kernels_2p20 pmc_kernels_2p22 \
         fq_bandwidth_2p20 fq_bandwidth_2p22 queue_diag ktime_probe callstats probe_pmc_kernels queue_probe_2p20 queue_probe_2p22 ubench_fpmul pytest_gpu_tail trip_probe; do
  [ -s $O/$f.txt ] && cp $O/$f.txt profiles/r6_$f.txt
done
for f in bench_line bench_line_2p16 bench_line_2p18 bench_line_2p22 bench_line_2p24 bench_line_small_memory; do [ -s $O/$f.json ] && cp $O/$f.json profiles/r6_$f.json; done
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/pmc_traffic.json
ls profiles/r6_* | wc -l

python - <<'PY' > profiles/r6_trip_budget.txt
import json, subprocess, sys
j = json.load(open("gpurun_out/r6prof/bench_line.json"))
sys.stdout.write(subprocess.run([sys.executable, "profiles/trip_budget.py", "gpurun_out/r6prof", "%.2f" % j["ms_per_step"], str(j["config"]["fs_trips_per_proof"])], capture_output=True, text=True).stdout)
PY
wc -l profiles/r6_trip_budget.txt
{ echo "# VERDICT r5 #2's file: the F_q kernels launch by launch at 2^20 and 2^22 (the two tables below are r6_fq_bandwidth_2p20.txt and _2p22.txt)"; echo; echo "#### 2^20"; cat profiles/r6_fq_bandwidth_2p20.txt; echo; echo "#### 2^22"; cat profiles/r6_fq_bandwidth_2p22.txt; } > profiles/r6_fq_bandwidth.txt
