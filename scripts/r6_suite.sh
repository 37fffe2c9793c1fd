mkdir -p gpurun_out/suite
timeout 3400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/suite/pytest_tail.txt; cat gpurun_out/suite/pytest_tail.txt
python bench.py > gpurun_out/suite/bench_line.json 2> gpurun_out/suite/bench_line.err; python - <<'PY'
import json; j=json.load(open("gpurun_out/suite/bench_line.json")); print(j["ms_per_step"], j["value"], j["config"].get("matches_oracle_digest"), j["config"].get("matches_oracle_live"), j["roofline"]["alu"]["frac"]); print(j["phases_ms"])
PY
