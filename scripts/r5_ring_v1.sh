mkdir -p gpurun_out/r5e
(SPARTAN_OPTIONS="testing.unlock=1,msm.form=2" timeout 600 python tests/msm_forms_worker.py 7 2>&1 | tail -5) > gpurun_out/r5e/ring_worker.txt
(timeout 300 python bench/msm_lds_probe.py 20 2>&1 | tail -12) > gpurun_out/r5e/probe_20.txt
cat gpurun_out/r5e/ring_worker.txt gpurun_out/r5e/probe_20.txt
bash scripts/gpu_ab.sh r5e 2 "wide:" "ring:msm.form=2" "ring6:msm.form=2,bg.eighths=6" 2>&1 | tail -12 | tee gpurun_out/r5e/ab.txt
