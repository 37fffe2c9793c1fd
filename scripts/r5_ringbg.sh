mkdir -p gpurun_out/r5j
bash scripts/gpu_ab.sh r5j 2 "wide:" "ringbg5:msm.form=4" "ringbg6:msm.form=4,bg.eighths=6" "ring6:msm.form=2,bg.eighths=6" 2>&1 | tail -10 | tee gpurun_out/r5j/ab.txt
