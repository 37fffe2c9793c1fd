mkdir -p gpurun_out/q3 spartan_amd/lib_prio
cp spartan_amd/lib/libspartan_hip_prio.so spartan_amd/lib_prio/libspartan_hip.so; cp spartan_amd/lib/libspartan_host.so spartan_amd/lib_prio/
AB_STEPS=20 bash scripts/gpu_ab.sh q3/ab20 2 "base:" "q5:msm.form=4" "q6:msm.form=4,bg.eighths=6" "q7:msm.form=4,bg.eighths=7" "q6w8:msm.form=4,bg.eighths=6,msm.q_bg_waves=8" \
   "Pq8w8@lib_prio:msm.form=4,bg.eighths=8,msm.q_bg_waves=8" "Pq8w12@lib_prio:msm.form=4,bg.eighths=8" "Pbase@lib_prio:" "Pq6@lib_prio:msm.form=4,bg.eighths=6" 2>&1 | tee gpurun_out/q3/ab20.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=400 bash scripts/gpu_ab.sh q3/ab22 1 "base:" "q5:msm.form=4" "q6:msm.form=4,bg.eighths=6" "q7:msm.form=4,bg.eighths=7" "Pq8w8@lib_prio:msm.form=4,bg.eighths=8,msm.q_bg_waves=8" "Pq8w12@lib_prio:msm.form=4,bg.eighths=8" 2>&1 | tee gpurun_out/q3/ab22.txt
