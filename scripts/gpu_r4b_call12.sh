#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4b12; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
./bench/ubench_fpmul > $O/ubench_new.txt 2>&1; tail -14 $O/ubench_new.txt
for rep in 1 2; do for lib in libspartan_hip.so libspartan_hip_oldshift.so; do echo "== $lib" >> $O/msm.txt; SPARTAN_HIP_LIB=$R/spartan_amd/lib/$lib SPARTAN_MSM_WBITS=14 timeout 300 python bench/msm_probe.py 2>&1 | tail -2 >> $O/msm.txt; done; done; cat $O/msm.txt
OS=$R/spartan_amd/lib/libspartan_hip_oldshift.so
bash scripts/gpu_ab.sh r4b12 3 "pkmov:" "oldshift:LD_PRELOAD=$OS,SPARTAN_HIP_LIB=$OS" > $O/ab_shift.txt 2>&1
cat $O/ab_shift.txt
