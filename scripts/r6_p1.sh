mkdir -p gpurun_out/p1
timeout 2400 python -m pytest tests/test_gpu_proofs.py tests/test_golden.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
AB_STEPS=20 bash scripts/gpu_ab.sh p1/ab20 3 "planned:" "perset:msm.plan_pair=0" 2>&1 | tee gpurun_out/p1/ab20.txt
AB_LOG2=24 AB_STEPS=3 AB_TIMEOUT=900 bash scripts/gpu_ab.sh p1/ab24 1 "planned:" "perset:msm.plan_pair=0" 2>&1 | tee gpurun_out/p1/ab24.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=400 bash scripts/gpu_ab.sh p1/ab22 1 "planned:" 2>&1 | tee gpurun_out/p1/ab22.txt
