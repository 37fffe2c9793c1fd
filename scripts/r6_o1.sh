mkdir -p gpurun_out/o1 spartan_amd/lib_occ
cp spartan_amd/lib/libspartan_hip_occ.so spartan_amd/lib_occ/libspartan_hip.so; cp spartan_amd/lib/libspartan_host.so spartan_amd/lib_occ/
AB_STEPS=20 bash scripts/gpu_ab.sh o1/ab20 3 "default:" "occ@lib_occ:" 2>&1 | tee gpurun_out/o1/ab20.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=400 bash scripts/gpu_ab.sh o1/ab22 3 "default:" "occ@lib_occ:" 2>&1 | tee gpurun_out/o1/ab22.txt
