mkdir -p gpurun_out/r5final
(timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "narrow_wide or row_msm_forms" 2>&1 | tail -5) | tee gpurun_out/r5final/pytest_forms.txt
bash profiles/collect_r5.sh > gpurun_out/r5final/collect.log 2>&1
tail -3 gpurun_out/r5final/collect.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r5final/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r5final/smoke.txt
