mkdir -p gpurun_out/q2
timeout 600 python bench/msm_queue_probe.py 20 12/2/64,12/2/128,12/2/32,8/2/64,8/2/128 w,c,h,d > gpurun_out/q2/probe20.txt 2>&1
cat gpurun_out/q2/probe20.txt
AB_STEPS=20 bash scripts/gpu_ab.sh q2/ab20 2 "base:" "q5:msm.form=4" "q6:msm.form=4,bg.eighths=6" "q7:msm.form=4,bg.eighths=7" "q5w8:msm.form=4,msm.q_bg_waves=8" "q8w8:msm.form=4,bg.eighths=8,msm.q_bg_waves=8" "q6u128:msm.form=4,bg.eighths=6,msm.q_units=128" 2>&1 | tee gpurun_out/q2/ab20.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=400 bash scripts/gpu_ab.sh q2/ab22 1 "base:" "q5:msm.form=4" "q6:msm.form=4,bg.eighths=6" "q8w8:msm.form=4,bg.eighths=8,msm.q_bg_waves=8" 2>&1 | tee gpurun_out/q2/ab22.txt
