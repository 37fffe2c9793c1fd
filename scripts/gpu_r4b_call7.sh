#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4b7; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "inner_product" > $O/pytest_ipa.txt 2>&1; echo "rc $?" >> $O/pytest_ipa.txt; tail -12 $O/pytest_ipa.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
bash scripts/gpu_ab.sh r4b7 3 "hostfin:" "devfin:SPARTAN_IPA_FINISH_DEVICE=1" > $O/ab_fin.txt 2>&1
cat $O/ab_fin.txt
