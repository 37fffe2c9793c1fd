mkdir -p gpurun_out/e1
timeout 1200 python -m pytest tests/test_gpu_proofs.py -m gpu -x -q 2>&1 | tail -5
AB_STEPS=20 bash scripts/gpu_ab.sh e1/ab20 3 "default:" "evalpass:polyeval.eval_from_opening=0" 2>&1 | tee gpurun_out/e1/ab20.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=400 bash scripts/gpu_ab.sh e1/ab22 2 "default:" "evalpass:polyeval.eval_from_opening=0" 2>&1 | tee gpurun_out/e1/ab22.txt
