mkdir -p gpurun_out/f1
AB_STEPS=20 bash scripts/gpu_ab.sh f1/ab20 3 "fit1:" "fit0:launch.fit=0" 2>&1 | tee gpurun_out/f1/ab20.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=400 bash scripts/gpu_ab.sh f1/ab22 2 "fit1:" "fit0:launch.fit=0" 2>&1 | tee gpurun_out/f1/ab22.txt
