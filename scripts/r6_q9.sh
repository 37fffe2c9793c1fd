mkdir -p gpurun_out/q9 spartan_amd/lib_noprio
cp spartan_amd/lib/libspartan_hip_noprio.so spartan_amd/lib_noprio/libspartan_hip.so; cp spartan_amd/lib/libspartan_host.so spartan_amd/lib_noprio/
SPARTAN_OPTIONS=testing.unlock=1,msm.form=4 timeout 900 python tests/msm_forms_worker.py 7 > gpurun_out/q9/forms_a.txt 2>&1; echo "rc $?" >> gpurun_out/q9/forms_a.txt
SPARTAN_OPTIONS=testing.unlock=1,msm.form=4,msm.q_coresident=0,msm.q_depth=3,msm.q_waves=8,msm.q_bg_waves=4,msm.q_units=4,msm.wbits=12,bg.eighths=6 timeout 900 python tests/msm_forms_worker.py 7 > gpurun_out/q9/forms_b.txt 2>&1; echo "rc $?" >> gpurun_out/q9/forms_b.txt
tail -n 2 gpurun_out/q9/forms_a.txt gpurun_out/q9/forms_b.txt
AB_STEPS=20 bash scripts/gpu_ab.sh q9/ab20 2 "base:" "Nbase@lib_noprio:" "qco:msm.form=4" "qco12:msm.form=4,msm.q_bg_waves=12" "qco_w15:msm.form=4,msm.wide_gb=200" "base_w15:msm.wide_gb=200" "qshare5:msm.form=4,msm.q_coresident=0,msm.q_bg_waves=12" "qco_u32:msm.form=4,msm.q_units=32" 2>&1 | tee gpurun_out/q9/ab20.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=400 bash scripts/gpu_ab.sh q9/ab22 2 "base:" "qco:msm.form=4" "qco12:msm.form=4,msm.q_bg_waves=12" 2>&1 | tee gpurun_out/q9/ab22.txt
