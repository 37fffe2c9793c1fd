mkdir -p gpurun_out/q4 spartan_amd/lib_prio
cp spartan_amd/lib/libspartan_hip_prio.so spartan_amd/lib_prio/libspartan_hip.so; cp spartan_amd/lib/libspartan_host.so spartan_amd/lib_prio/
SPARTAN_OPTIONS=testing.unlock=1,msm.form=4 timeout 900 python tests/msm_forms_worker.py 7 > gpurun_out/q4/forms_a.txt 2>&1; echo "rc $?" >> gpurun_out/q4/forms_a.txt
SPARTAN_OPTIONS=testing.unlock=1,msm.form=4,msm.q_depth=3,msm.q_waves=8,msm.q_bg_waves=4,msm.q_units=4,msm.wbits=12,bg.eighths=6 timeout 900 python tests/msm_forms_worker.py 7 > gpurun_out/q4/forms_b.txt 2>&1; echo "rc $?" >> gpurun_out/q4/forms_b.txt
tail -n 3 gpurun_out/q4/forms_a.txt gpurun_out/q4/forms_b.txt
timeout 600 python bench/msm_queue_probe.py 20 12/2/16,12/2/32,12/2/64,12/2/128,8/2/32,8/3/32 w,h,d > gpurun_out/q4/probe20.txt 2>&1
cat gpurun_out/q4/probe20.txt
timeout 900 python bench/msm_queue_probe.py 22 12/2/32,12/2/64,8/2/32 w,h > gpurun_out/q4/probe22.txt 2>&1
cat gpurun_out/q4/probe22.txt
AB_STEPS=20 bash scripts/gpu_ab.sh q4/ab20 2 "base:" "q5:msm.form=4" "q6:msm.form=4,bg.eighths=6" "q5u32:msm.form=4,msm.q_units=32" "q6u32:msm.form=4,bg.eighths=6,msm.q_units=32" \
   "Pq8w8@lib_prio:msm.form=4,bg.eighths=8,msm.q_bg_waves=8" "Pbase@lib_prio:" 2>&1 | tee gpurun_out/q4/ab20.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=400 bash scripts/gpu_ab.sh q4/ab22 1 "base:" "q5:msm.form=4" "q6:msm.form=4,bg.eighths=6" "q7:msm.form=4,bg.eighths=7" "Pq8w8@lib_prio:msm.form=4,bg.eighths=8,msm.q_bg_waves=8" 2>&1 | tee gpurun_out/q4/ab22.txt
