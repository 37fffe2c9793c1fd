mkdir -p gpurun_out/r5c
(SPARTAN_OPTIONS="testing.unlock=1,msm.lds_bits=10,msm.form=1" timeout 600 python tests/msm_forms_worker.py 7 2>&1 | tail -5) > gpurun_out/r5c/lds_worker.txt
(timeout 300 python bench/msm_lds_probe.py 20 2>&1 | tail -12) > gpurun_out/r5c/lds_probe_20.txt
(SPARTAN_HIP_LIB=$PWD/spartan_amd/lib/libspartan_hip_ldsdiag.so timeout 300 python bench/msm_lds_probe.py 20 10 diag 2>&1 | grep "derefs whole\|witness") > gpurun_out/r5c/lds_diag.txt
cat gpurun_out/r5c/lds_worker.txt gpurun_out/r5c/lds_probe_20.txt gpurun_out/r5c/lds_diag.txt
