#!/bin/bash
# round 4, GPU call 3: the new defaults (aligned entries, balanced foreground MSM, strip-form background) against their alternatives; full GPU suite
R=$(pwd); O=$R/gpurun_out/r4c3; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
P=$R/spartan_amd/lib/libspartan_hip_packed96.so
bash scripts/gpu_ab.sh r4c3 3 "default:" "flat0:SPARTAN_MSM_FLAT=0" "flatbg:SPARTAN_MSM_FLAT_BG=1" "flat1:SPARTAN_MSM_FLAT=1" "packed96_flat0:LD_PRELOAD=$P,SPARTAN_HIP_LIB=$P,SPARTAN_MSM_FLAT=0" "w15:SPARTAN_MSM_WIDE_GB=200,SPARTAN_MSM_TABLE_GB=200" > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
j=json.load(open("gpurun_out/r4c3/bench_default.json"))
print(j["ms_per_step"], j.get("snark_encode"), j.get("nizk_prove"), j.get("throughput_concurrent"))
PY
