mkdir -p gpurun_out/r5g
timeout 600 python -m pytest tests/test_gpu_sanitizers.py -x -q 2>&1 | tail -15 | tee gpurun_out/r5g/tsan_pytest.txt
cp gpurun_out/tsan_report.txt gpurun_out/r5g/ 2>/dev/null; grep -c "WARNING: ThreadSanitizer" gpurun_out/tsan_report.txt; head -c 3000 gpurun_out/tsan_report.txt
