mkdir -p gpurun_out/suite
timeout 3400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/suite/pytest_tail.txt; cat gpurun_out/suite/pytest_tail.txt
