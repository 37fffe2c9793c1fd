#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4b6; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "product_tree" > $O/pytest_k.txt 2>&1; echo "rc $?" >> $O/pytest_k.txt; tail -12 $O/pytest_k.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
bash scripts/gpu_ab.sh r4b6 3 "fused:" "separate:SPARTAN_NO_HASH_FUSE=1" > $O/ab_hash.txt 2>&1
cat $O/ab_hash.txt
