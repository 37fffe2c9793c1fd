#!/bin/bash
# launch-ahead trips: correctness first (A/B switch test incl. the give-up hook, sum-check kernel tests), then interleaved A/B at 2^20
mkdir -p gpurun_out/la
timeout 1500 python -m pytest tests/test_gpu_proofs.py -x -q -m gpu -k "every_ab_switch or full_size_properties or golden or caller_owned" 2>&1 | grep -v "RCCL\|NCCL\|^$" | tail -30 > gpurun_out/la/pytest.txt
tail -5 gpurun_out/la/pytest.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_large.py -x -q -m gpu -k "sumcheck or cubic or round" 2>&1 | grep -E "passed|failed" | tail -3
bash scripts/gpu_ab.sh la 3 "ahead:" "off:sumcheck.launch_ahead=0" > gpurun_out/la/ab20.txt 2>&1
cat gpurun_out/la/ab20.txt
