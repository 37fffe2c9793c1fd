#!/bin/bash
# round 4, GPU call 1: gather ceiling, parity of the balanced MSM + DPP tree, A/B of both on the 2^20 proof
R=$(pwd); O=$R/gpurun_out/r4c1; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 ./bench/gather_probe > $O/gather_probe.txt 2>&1; echo "rc $?" >> $O/gather_probe.txt ) 
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
# MSM launch alone at each form (14-bit: the derefs stream)
for f in 0 1 2; do
  echo "== SPARTAN_MSM_FLAT=$f" >> $O/msm_probe.txt
  SPARTAN_MSM_FLAT=$f SPARTAN_MSM_WBITS=14 timeout 300 python bench/msm_probe.py 2>&1 | tail -3 >> $O/msm_probe.txt
done
echo "== SPARTAN_MSM_FLAT=2 rounds 2" >> $O/msm_probe.txt
SPARTAN_MSM_FLAT=2 SPARTAN_MSM_FLAT_ROUNDS=2 SPARTAN_MSM_WBITS=14 timeout 300 python bench/msm_probe.py 2>&1 | tail -3 >> $O/msm_probe.txt
cat $O/msm_probe.txt
T=$R/spartan_amd/lib/libspartan_hip_treelds.so
bash scripts/gpu_ab.sh r4c1 2 "old:LD_PRELOAD=$T,SPARTAN_HIP_LIB=$T,SPARTAN_MSM_FLAT=0" "flat2_treelds:LD_PRELOAD=$T,SPARTAN_HIP_LIB=$T" "flat0_dpp:SPARTAN_MSM_FLAT=0" "flat1_dpp:SPARTAN_MSM_FLAT=1" "flat2_dpp:" "flat2r2_dpp:SPARTAN_MSM_FLAT_ROUNDS=2" > $O/ab.txt 2>&1
cat $O/ab.txt
head -60 $O/gather_probe.txt
