#!/bin/bash
# round 4, GPU call 4: shard tests (batched cubic sum-checks, chunked evaluations), full suite, the bench line with its new fields
R=$(pwd); O=$R/gpurun_out/r4c4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shard.py -m gpu -x -q > $O/pytest_shard.txt 2>&1; echo "rc $?" >> $O/pytest_shard.txt; tail -25 $O/pytest_shard.txt
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_shard.py > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt; tail -5 $O/bench_default.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r4c4/bench_default.json"))
print(j["ms_per_step"], j["config"].get("matches_oracle_digest"), j["config"].get("matches_oracle_live"))
print(json.dumps(j["roofline"]["alu"], indent=0)[:3000])
print(json.dumps(j.get("cpu_baseline"))[:600])
print(json.dumps(j.get("cpu_baseline_one_core"))[:400])
print(j.get("snark_encode"), j.get("phases_ms"))
PY
