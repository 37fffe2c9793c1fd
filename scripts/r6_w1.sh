mkdir -p gpurun_out/w1
timeout 3400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/w1/pytest_tail.txt; cat gpurun_out/w1/pytest_tail.txt
AB_STEPS=20 bash scripts/gpu_ab.sh w1/ab20 2 "default:" "w19:msm.windows=19" "w18:msm.windows=18" "w17:msm.windows=17,msm.wide_gb=200" 2>&1 | tee gpurun_out/w1/ab20.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=400 bash scripts/gpu_ab.sh w1/ab22 2 "default:" "w19:msm.windows=19" 2>&1 | tee gpurun_out/w1/ab22.txt
AB_LOG2=24 AB_STEPS=3 AB_TIMEOUT=900 bash scripts/gpu_ab.sh w1/ab24 1 "default:" 2>&1 | tee gpurun_out/w1/ab24.txt
