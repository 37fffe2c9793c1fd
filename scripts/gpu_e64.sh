#!/bin/bash
# timing experiment: 64-byte table entries in the row MSM (make -C spartan_amd/csrc e64 builds libspartan_hip_e64.so, which computes wrong points; only the times mean anything)
R=$(pwd); O=$R/gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
for b in 14 15; do
  for lib in libspartan_hip.so libspartan_hip_e64.so; do
    echo "== $lib wbits $b" >> $O/e64.txt
    SPARTAN_HIP_LIB=$R/spartan_amd/lib/$lib SPARTAN_MSM_WBITS=$b timeout 300 python bench/msm_probe.py 2>&1 | tail -2 >> $O/e64.txt
  done
done
done
cat $O/e64.txt
