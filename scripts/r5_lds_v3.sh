mkdir -p gpurun_out/r5d
(SPARTAN_OPTIONS="testing.unlock=1,msm.lds_bits=10,msm.form=1" timeout 600 python tests/msm_forms_worker.py 7 2>&1 | tail -5) > gpurun_out/r5d/lds_worker.txt
(timeout 300 python bench/msm_lds_probe.py 20 2>&1 | tail -12) > gpurun_out/r5d/lds_probe_20.txt
(SPARTAN_HIP_LIB=$PWD/spartan_amd/lib/libspartan_hip_ldsdiag.so timeout 300 python bench/msm_lds_probe.py 20 10 diag 2>&1 | grep "derefs whole\|derefs half") > gpurun_out/r5d/lds_diag.txt
cat gpurun_out/r5d/lds_worker.txt gpurun_out/r5d/lds_probe_20.txt gpurun_out/r5d/lds_diag.txt
bash scripts/gpu_ab.sh r5d 2 "wide:" "lds:msm.lds_bits=10,msm.form=1" "lds_small:msm.lds_bits=10,msm.form=1,msm.wbits=10" 2>&1 | tail -12 | tee gpurun_out/r5d/ab.txt
