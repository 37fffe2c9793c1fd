mkdir -p gpurun_out/q7
L=$(pwd)/spartan_amd/lib
echo "== SP_Q_DIAG=4 (gathers issued, not waited for)" >> gpurun_out/q7/diag.txt
PROBE_NOCHECK=1 SPARTAN_HIP_LIB=$L/libspartan_hip_qdiag4.so timeout 600 python bench/msm_queue_probe.py 22 12/2/64,8/3/64,8/2/64 h >> gpurun_out/q7/diag.txt 2>&1
cat gpurun_out/q7/diag.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OLDPWD/gpurun_out/q7/pmc_wait -- python $OLDPWD/bench/msm_queue_probe.py 22 12/2/64,8/3/64 h > $OLDPWD/gpurun_out/q7/pmc_wait.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OLDPWD/gpurun_out/q7/pmc_busy -- python $OLDPWD/bench/msm_queue_probe.py 22 12/2/64,8/3/64 h > $OLDPWD/gpurun_out/q7/pmc_busy.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OLDPWD/gpurun_out/q7/pmc_lds -- python $OLDPWD/bench/msm_queue_probe.py 22 12/2/64,8/3/64 h > $OLDPWD/gpurun_out/q7/pmc_lds.log 2>&1
cd $OLDPWD
for d in pmc_wait pmc_busy pmc_lds; do f=$(find gpurun_out/q7/$d -name "*counter_collection.csv" | head -1); python profiles/pmc_counters.py $f --kernels k_msm_q,k_msm_rows,k_msm_ring > gpurun_out/q7/$d.txt 2>&1; rm -rf gpurun_out/q7/$d; done
cat gpurun_out/q7/pmc_wait.txt gpurun_out/q7/pmc_busy.txt gpurun_out/q7/pmc_lds.txt
