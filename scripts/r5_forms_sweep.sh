mkdir -p gpurun_out/r5f
bash scripts/gpu_ab.sh r5f 2 "wide:" "ring6:msm.form=2,bg.eighths=6" "ring7:msm.form=2,bg.eighths=7" "lds6:msm.lds_bits=10,msm.form=1,bg.eighths=6" "lds7:msm.lds_bits=10,msm.form=1,bg.eighths=7" "wide6:bg.eighths=6" 2>&1 | tail -14 | tee gpurun_out/r5f/ab_2p20.txt
(timeout 600 python bench/msm_lds_probe.py 24 2>&1 | tail -12) > gpurun_out/r5f/probe_24.txt; cat gpurun_out/r5f/probe_24.txt
AB_LOG2=24 AB_STEPS=5 AB_TIMEOUT=600 bash scripts/gpu_ab.sh r5f24 1 "wide:" "ring:msm.form=2" "lds:msm.lds_bits=10,msm.form=1" 2>&1 | tail -8 | tee gpurun_out/r5f/ab_2p24.txt
