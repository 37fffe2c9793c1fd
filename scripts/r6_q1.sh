mkdir -p gpurun_out/q1
SPARTAN_OPTIONS=testing.unlock=1,msm.form=4 timeout 900 python tests/msm_forms_worker.py 7 > gpurun_out/q1/forms_a.txt 2>&1; echo "rc $?" >> gpurun_out/q1/forms_a.txt
SPARTAN_OPTIONS=testing.unlock=1,msm.form=4,msm.q_depth=3,msm.q_waves=8,msm.q_bg_waves=4,msm.q_units=4,msm.wbits=12,bg.eighths=6 timeout 900 python tests/msm_forms_worker.py 7 > gpurun_out/q1/forms_b.txt 2>&1; echo "rc $?" >> gpurun_out/q1/forms_b.txt
tail -n 3 gpurun_out/q1/forms_a.txt gpurun_out/q1/forms_b.txt
timeout 600 python bench/msm_queue_probe.py 20 12/2/64,8/3/64,8/2/64,4/3/64,12/2/32,12/2/128 w,c,h,d > gpurun_out/q1/probe20.txt 2>&1
cat gpurun_out/q1/probe20.txt
timeout 900 python bench/msm_queue_probe.py 22 12/2/64,8/3/64,12/2/128 w,h > gpurun_out/q1/probe22.txt 2>&1
cat gpurun_out/q1/probe22.txt
