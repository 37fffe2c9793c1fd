mkdir -p gpurun_out/q6
L=$(pwd)/spartan_amd/lib
SPARTAN_OPTIONS=testing.unlock=1,msm.form=4 timeout 900 python tests/msm_forms_worker.py 7 > gpurun_out/q6/forms_a.txt 2>&1; echo "rc $?" >> gpurun_out/q6/forms_a.txt
SPARTAN_OPTIONS=testing.unlock=1,msm.form=4,msm.q_depth=3,msm.q_waves=8,msm.q_bg_waves=4,msm.q_units=4,msm.wbits=12,bg.eighths=6 timeout 900 python tests/msm_forms_worker.py 7 > gpurun_out/q6/forms_b.txt 2>&1; echo "rc $?" >> gpurun_out/q6/forms_b.txt
tail -n 3 gpurun_out/q6/forms_a.txt gpurun_out/q6/forms_b.txt
timeout 600 python bench/msm_queue_probe.py 20 12/2/64,8/2/64,8/3/64 w,h,d > gpurun_out/q6/probe20.txt 2>&1; cat gpurun_out/q6/probe20.txt
timeout 600 python bench/msm_queue_probe.py 22 12/2/64,8/2/64,8/3/64 w,h > gpurun_out/q6/probe22.txt 2>&1; cat gpurun_out/q6/probe22.txt
for n in 1 2; do
  echo "== SP_Q_DIAG=$n (1: no gathers, 2: no additions)" >> gpurun_out/q6/diag.txt
  PROBE_NOCHECK=1 SPARTAN_HIP_LIB=$L/libspartan_hip_qdiag$n.so timeout 600 python bench/msm_queue_probe.py 22 12/2/64,8/3/64 h >> gpurun_out/q6/diag.txt 2>&1
done
cat gpurun_out/q6/diag.txt
