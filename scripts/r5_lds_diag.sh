mkdir -p gpurun_out/r5b
(SPARTAN_HIP_LIB=$PWD/spartan_amd/lib/libspartan_hip_ldsdiag.so timeout 300 python bench/msm_lds_probe.py 20 10 diag 2>&1 | tail -40) > gpurun_out/r5b/lds_diag.txt
cat gpurun_out/r5b/lds_diag.txt
