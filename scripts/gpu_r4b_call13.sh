#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4b13; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_large.py tests/test_golden.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt
OS=$R/spartan_amd/lib/libspartan_hip_plainmadd.so
bash scripts/gpu_ab.sh r4b13 4 "msmpk:" "plain:LD_PRELOAD=$OS,SPARTAN_HIP_LIB=$OS" > $O/ab_msmpk.txt 2>&1
cat $O/ab_msmpk.txt
