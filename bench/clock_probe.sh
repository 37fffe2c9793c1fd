#!/bin/bash
# Scratch probe: GPU clocks while the latency-bound prover runs
(for i in $(seq 1 12); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -3 | tr '\n' ' '; echo; sleep 0.5; done) > gpurun_out/clocks.txt &
BENCH_NO_PROF=1 python bench.py --steps 40 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
wait
cat gpurun_out/clocks.txt
rocm-smi --showperflevel 2>/dev/null | head -8
