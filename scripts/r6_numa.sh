#!/bin/bash
# option host.pin_thread (sp_ctx_create narrows the proving thread's affinity to the GPU's NUMA node) x sumcheck.launch_ahead, interleaved
python - <<'PY'
import os
from spartan_amd import capi
print("affinity before ctx:", len(os.sched_getaffinity(0)))
c = capi.Ctx(0)
print("affinity after ctx :", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:2], "...")
c.close()
PY
bash scripts/gpu_ab.sh numa 5 "pin_ahead:" "pin_off:sumcheck.launch_ahead=0" "nopin_ahead:host.pin_thread=0" "nopin_off:host.pin_thread=0,sumcheck.launch_ahead=0" 2>&1 | grep -v phases
python - <<'PY'
import json,glob
for n in ("pin_ahead","pin_off","nopin_ahead","nopin_off"):
    v=[json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("gpurun_out/numa/ab_%s_*.json"%n))]
    print(n, [round(x,2) for x in v])
PY
