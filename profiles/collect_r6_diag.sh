#!/bin/bash
# Round 6: what each side of a tile of the queue-form row MSM costs, and what clock the chip holds while it runs (variant builds of the library:
# make -C spartan_amd/csrc variant NAME=qdiagN FLAGS=-DSP_Q_DIAG=N — WRONG RESULTS by construction for N & 7, never in the product), the
# in-kernel stamps of the two latency kernels (make ktime), and the per-entry-point wall time of one proof (option host.callstats).
R=$(pwd); O=$R/gpurun_out/r6prof; mkdir -p $O; L=$R/spartan_amd/lib
: > $O/queue_diag.txt
for n in 1 2 4 8 9; do
  case $n in 1) what="no gathers: every addition takes the neutral entry (the ALU side alone)";; 2) what="no additions: the gathers and their waits alone";;
    4) what="gathers issued, never waited for (the cost of ISSUING them)";; 8) what="the full kernel + its shader clock";; 9) what="no gathers + its shader clock";; esac
  echo "== SP_Q_DIAG=$n: $what" >> $O/queue_diag.txt
  PROBE_NOCHECK=1 SPARTAN_HIP_LIB=$L/libspartan_hip_qdiag$n.so timeout 600 python bench/msm_queue_probe.py 22 12/2/32,8/2/32 h 2>&1 | grep -v "^2^22 derefs half.*strip" | sort | uniq -c | sed 's/^ *1 //' >> $O/queue_diag.txt
done
cat $O/queue_diag.txt
timeout 600 python bench/ktime_probe.py > $O/ktime_probe.txt 2>&1; tail -30 $O/ktime_probe.txt
timeout 200 ./bench/trip_probe 1500 > $O/trip_probe.txt 2>&1; cat $O/trip_probe.txt
