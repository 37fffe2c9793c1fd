mkdir -p gpurun_out/e2
timeout 900 python -m pytest tests/test_gpu_proofs.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
AB_STEPS=20 bash scripts/gpu_ab.sh e2/ab20 3 "default:" 2>&1 | tee gpurun_out/e2/ab20.txt
AB_LOG2=22 AB_STEPS=8 AB_TIMEOUT=400 bash scripts/gpu_ab.sh e2/ab22 2 "default:" 2>&1 | tee gpurun_out/e2/ab22.txt
export TMPDIR=/tmp BENCH_NO_GATHER_PROBE=1; R=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/e2/st -- python $R/bench.py --log2-cons 22 --no-cpu-baseline --concurrent 0 --steps 2 --warmup 1 --no-side-metrics --no-strong > $R/gpurun_out/e2/st.log 2>&1 )
python profiles/summarize.py "$(find gpurun_out/e2/st -name '*_results.db' | head -1)" | grep -E "k_pt_encode|k_msm_reduce|k_msm_q" ; rm -rf gpurun_out/e2/st
