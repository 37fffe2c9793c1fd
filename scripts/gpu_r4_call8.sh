#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4c8; mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r4c8 3 "default:" "nothread:SPARTAN_NO_UPLOAD_THREAD=1" > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
