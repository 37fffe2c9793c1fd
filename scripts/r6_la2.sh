#!/bin/bash
mkdir -p gpurun_out/la
SP_AHEAD_TRACE=1 timeout 300 python bench.py --no-cpu-baseline --concurrent 0 --steps 3 --warmup 1 --no-side-metrics --no-strong > gpurun_out/la/trace20.json 2> gpurun_out/la/trace20.err
grep -c "arm seq" gpurun_out/la/trace20.err; grep -c "cancel" gpurun_out/la/trace20.err; grep -c "GAVE UP" gpurun_out/la/trace20.err; grep "rung trips" gpurun_out/la/trace20.err
for v in 1 0; do
SPARTAN_OPTIONS=testing.unlock=1,host.callstats=1,sumcheck.launch_ahead=$v timeout 300 python bench.py --no-cpu-baseline --concurrent 0 --steps 4 --warmup 1 --no-side-metrics --no-strong > gpurun_out/la/cs$v.json 2> gpurun_out/la/cs$v.err; grep callstats gpurun_out/la/cs$v.err | tail -44 | grep "bind2\|coeffs"
done
timeout 900 python -m pytest tests/test_gpu_proofs.py -x -q -m gpu -k "every_ab_switch" 2>&1 | grep -E "passed|failed" | tail -3
bash scripts/gpu_ab.sh la 3 "ahead:" "off:sumcheck.launch_ahead=0" > gpurun_out/la/ab20.txt 2>&1
cat gpurun_out/la/ab20.txt
