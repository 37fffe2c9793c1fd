mkdir -p gpurun_out/r5k
bash scripts/gpu_ab.sh r5k 3 "wide:" "ringbg5:msm.form=4" 2>&1 | tail -5 | tee gpurun_out/r5k/ab_2p20.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=600 bash scripts/gpu_ab.sh r5k22 2 "wide:" "ringbg5:msm.form=4" 2>&1 | tail -5 | tee gpurun_out/r5k/ab_2p22.txt
AB_LOG2=24 AB_STEPS=5 AB_TIMEOUT=600 bash scripts/gpu_ab.sh r5k24 1 "wide:" "ringbg5:msm.form=4" 2>&1 | tail -5 | tee gpurun_out/r5k/ab_2p24.txt
AB_LOG2=16 bash scripts/gpu_ab.sh r5k16 2 "wide:" "ringbg5:msm.form=4" 2>&1 | tail -5 | tee gpurun_out/r5k/ab_2p16.txt
