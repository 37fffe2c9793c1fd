mkdir -p gpurun_out/r5a
(SPARTAN_OPTIONS="testing.unlock=1,msm.lds_bits=10,msm.form=1" timeout 600 python tests/msm_forms_worker.py 7 2>&1 | tail -5) > gpurun_out/r5a/lds_worker.txt
(timeout 600 python bench/msm_lds_probe.py 20 2>&1 | tail -12) > gpurun_out/r5a/lds_probe_20.txt
cat gpurun_out/r5a/lds_worker.txt gpurun_out/r5a/lds_probe_20.txt
