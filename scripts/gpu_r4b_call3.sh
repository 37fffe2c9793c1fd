#!/bin/bash
# dedicated-addition tree in k_ipa_round: parity (IPA tests, proofs, golden), in-kernel stamps, A/B against the unified tree
R=$(pwd); O=$R/gpurun_out/r4b3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "inner_product" > $O/pytest_ipa.txt 2>&1; echo "rc $?" >> $O/pytest_ipa.txt; tail -15 $O/pytest_ipa.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
SPARTAN_KTIME=1 timeout 300 python bench/ktime_probe.py > $O/ktime_ded.txt 2>&1; grep -A12 "k_ipa_round, n = 4096" $O/ktime_ded.txt | head -16
SPARTAN_IPA_UNIFIED_TREE=1 SPARTAN_KTIME=1 timeout 300 python bench/ktime_probe.py > $O/ktime_unified.txt 2>&1; grep -A12 "k_ipa_round, n = 4096" $O/ktime_unified.txt | head -16
bash scripts/gpu_ab.sh r4b3 3 "ded:" "unified:SPARTAN_IPA_UNIFIED_TREE=1" > $O/ab_tree.txt 2>&1
cat $O/ab_tree.txt
