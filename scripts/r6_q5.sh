mkdir -p gpurun_out/q5
L=$(pwd)/spartan_amd/lib
for n in 1 2; do
  echo "== SP_Q_DIAG=$n (1: no gathers, 2: no additions)" >> gpurun_out/q5/diag.txt
  PROBE_NOCHECK=1 SPARTAN_HIP_LIB=$L/libspartan_hip_qdiag$n.so timeout 600 python bench/msm_queue_probe.py 20 12/2/64,8/2/64,8/3/64 h >> gpurun_out/q5/diag.txt 2>&1
  PROBE_NOCHECK=1 SPARTAN_HIP_LIB=$L/libspartan_hip_qdiag$n.so timeout 600 python bench/msm_queue_probe.py 22 12/2/64,8/3/64 h >> gpurun_out/q5/diag.txt 2>&1
done
cat gpurun_out/q5/diag.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OLDPWD/gpurun_out/q5/pmc_wait -- python $OLDPWD/bench/msm_queue_probe.py 22 12/2/64,8/3/64 h > $OLDPWD/gpurun_out/q5/pmc_wait.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OLDPWD/gpurun_out/q5/pmc_busy -- python $OLDPWD/bench/msm_queue_probe.py 22 12/2/64,8/3/64 h > $OLDPWD/gpurun_out/q5/pmc_busy.log 2>&1
cd $OLDPWD
for d in pmc_wait pmc_busy; do f=$(find gpurun_out/q5/$d -name "*counter_collection.csv" | head -1); python profiles/pmc_counters.py $f > gpurun_out/q5/$d.txt 2>&1; rm -rf gpurun_out/q5/$d; done
cat gpurun_out/q5/pmc_wait.txt gpurun_out/q5/pmc_busy.txt
