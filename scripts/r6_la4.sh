#!/bin/bash
# launch-ahead with the main-stream wrapper (any use of the stream cancels a waiting kernel): how many are rung / cancelled in a proof, suite, A/B
mkdir -p gpurun_out/la
SPARTAN_OPTIONS=testing.unlock=1,host.callstats=1 timeout 300 python bench.py --no-cpu-baseline --concurrent 0 --steps 4 --warmup 1 --no-side-metrics --no-strong > /dev/null 2> gpurun_out/la/cs_final.err; grep "enqueued ahead" gpurun_out/la/cs_final.err
SPARTAN_OPTIONS=testing.unlock=1,host.callstats=1 timeout 300 python bench.py --log2-cons 22 --no-cpu-baseline --concurrent 0 --steps 2 --warmup 1 --no-side-metrics --no-strong > /dev/null 2> gpurun_out/la/cs_final22.err; grep "enqueued ahead" gpurun_out/la/cs_final22.err
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
bash scripts/gpu_ab.sh la4 3 "ahead:" "off:sumcheck.launch_ahead=0" 2>&1 | grep -v phases
