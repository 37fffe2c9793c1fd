#!/bin/bash
# the spread from one process to the next sits in the latency chain (evalproof 12.8 vs 13.7 ms): SMT siblings? one hardware thread per core of the GPU's node
# against the whole node (option host.pin_thread today) against no pinning
python - <<'PY'
import os
from spartan_amd import capi
c = capi.Ctx(0); m = sorted(os.sched_getaffinity(0)); print("pinned to", m[0], "...", m[-1], len(m)); c.close()
print(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % m[0]).read().strip())
PY
NODE=$(python - <<'PY'
import os
from spartan_amd import capi
c = capi.Ctx(0); m = sorted(os.sched_getaffinity(0)); c.close()
first = sorted({min(int(x) for x in open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % k).read().strip().replace("-", ",").split(",")) for k in m})
print(",".join(str(x) for x in first))
PY
)
echo "one thread per core: $(echo $NODE | cut -c1-60)..."
Q="--no-cpu-baseline --concurrent 0 --steps 30 --warmup 2 --no-side-metrics --no-strong"
run() { local name=$1; shift; BENCH_NO_GATHER_PROBE=1 "$@" python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3), 'resident', round(d['config']['resident_assignment']['ms_per_step'],3), 'spark', d['phases_ms']['evalproof_layered_network'])"; }
for rep in 1 2 3 4 5 6; do
  run "node (default)   " env
  run "one thread / core" taskset -c $NODE
  run "unpinned         " env SPARTAN_OPTIONS=host.pin_thread=0
done
