mkdir -p gpurun_out/q8
L=$(pwd)/spartan_amd/lib
for n in 8 9; do
echo "== SP_Q_DIAG=$n (8: full kernel + clock, 9: no gathers + clock)" >> gpurun_out/q8/diag.txt
PROBE_NOCHECK=1 SPARTAN_HIP_LIB=$L/libspartan_hip_qdiag$n.so timeout 600 python bench/msm_queue_probe.py 22 12/2/64 h 2>&1 | sort | uniq -c | sort -rn | head -12 >> gpurun_out/q8/diag.txt
done
cat gpurun_out/q8/diag.txt
