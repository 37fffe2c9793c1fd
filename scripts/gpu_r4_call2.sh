#!/bin/bash
# round 4, GPU call 2: ALU ceiling at the MSM's occupancy, 128-byte-aligned table entries, background share under the balanced MSM
R=$(pwd); O=$R/gpurun_out/r4c2; mkdir -p $O
export TMPDIR=/tmp
./bench/ubench_fpmul --json > $O/ubench.json 2>&1; cat $O/ubench.json
timeout 600 python -m pytest tests/test_gpu_shard.py -m gpu -x -q > $O/pytest_shard.txt 2>&1; tail -3 $O/pytest_shard.txt
A=$R/spartan_amd/lib/libspartan_hip_a128.so
for b in 14 15; do for f in 0 2; do
  echo "== default lib wbits $b flat $f" >> $O/msm_probe.txt
  SPARTAN_MSM_FLAT=$f SPARTAN_MSM_WBITS=$b timeout 300 python bench/msm_probe.py 2>&1 | tail -2 >> $O/msm_probe.txt
  echo "== a128 lib wbits $b flat $f" >> $O/msm_probe.txt
  SPARTAN_HIP_LIB=$A SPARTAN_MSM_FLAT=$f SPARTAN_MSM_WBITS=$b timeout 300 python bench/msm_probe.py 2>&1 | tail -2 >> $O/msm_probe.txt
done; done
cat $O/msm_probe.txt
bash scripts/gpu_ab.sh r4c2 2 "flat0_bg5:SPARTAN_MSM_FLAT=0" "flat2_bg5:" "flat2_bg4:SPARTAN_BG_EIGHTHS=4" "flat2_bg3:SPARTAN_BG_EIGHTHS=3" "flat2_bg6:SPARTAN_BG_EIGHTHS=6" "flat2_noov:SPARTAN_NO_OVERLAP=1" \
  "a128_flat2_bg5:LD_PRELOAD=$A,SPARTAN_HIP_LIB=$A" "a128_flat2_bg4:LD_PRELOAD=$A,SPARTAN_HIP_LIB=$A,SPARTAN_BG_EIGHTHS=4" "a128_flat0_bg5:LD_PRELOAD=$A,SPARTAN_HIP_LIB=$A,SPARTAN_MSM_FLAT=0" \
  "a128_w15_flat2_bg4:LD_PRELOAD=$A,SPARTAN_HIP_LIB=$A,SPARTAN_BG_EIGHTHS=4,SPARTAN_MSM_WIDE_GB=200,SPARTAN_MSM_TABLE_GB=200" "w15_flat2_bg4:SPARTAN_BG_EIGHTHS=4,SPARTAN_MSM_WIDE_GB=200" > $O/ab.txt 2>&1
cat $O/ab.txt
