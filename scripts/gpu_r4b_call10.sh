#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4b10; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_large.py tests/test_golden.py tests/test_gpu_proofs.py tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
SPARTAN_KTIME=1 timeout 300 python bench/ktime_probe.py > $O/ktime.txt 2>&1; grep -A12 "k_ipa_round, n = 4096" $O/ktime.txt | head -14
bash scripts/gpu_ab.sh r4b10 3 "helpers:" "nohelpers:SPARTAN_NO_COMMIT_HELPERS=1" > $O/ab_helpers.txt 2>&1
cat $O/ab_helpers.txt
