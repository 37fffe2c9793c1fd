mkdir -p gpurun_out/r3a
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/r3a/tests.log
timeout 300 python bench.py --steps 20 --warmup 2 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
SPARTAN_IPA_UNFUSED=1 timeout 200 python bench.py --steps 10 --no-cpu-baseline --no-side-metrics --concurrent 0 > gpurun_out/r3a/bench_ipa_unfused.json 2> gpurun_out/r3a/bench_ipa_unfused.err
SPARTAN_MSM_WIDE_GB=200 timeout 200 python bench.py --steps 10 --no-cpu-baseline --no-side-metrics --concurrent 0 > gpurun_out/r3a/bench_w15.json 2> gpurun_out/r3a/bench_w15.err
SPARTAN_CALLSTATS=1 timeout 200 python bench.py --steps 3 --no-cpu-baseline --no-side-metrics --concurrent 0 > gpurun_out/r3a/bench_callstats.json 2> gpurun_out/r3a/callstats.err
tail -3 gpurun_out/r3a/tests.log
python - <<'PY'
import json
for f in ("bench","bench_ipa_unfused","bench_w15"):
    try:
        d=json.load(open(f"gpurun_out/r3a/{f}.json"))
        print(f, round(d["ms_per_step"],3), d["config"].get("matches_oracle_digest"), d["config"]["resident_assignment"]["ms_per_step"], d["config"]["fs_trips_per_proof"], d["config"]["table_GB"], d["phases_ms"])
    except Exception as e: print(f, "ERR", e)
PY
