#!/bin/bash
# A/B harness: `bash scripts/gpu_ab.sh OUTDIR REPS "NAME1:opt.a=1,opt.b=0" "NAME2:" ...` runs bench.py (headline step only) REPS times per
# variant, interleaved, each with its library options (spartan_amd/csrc/options.hpp) handed over through SPARTAN_OPTIONS, and prints the
# per-variant ms/step (min / median) — box-to-box and run-to-run noise is +-0.5 ms per proof. AB_LOG2 (default 20) picks the size.
O=gpurun_out/$1; REPS=$2; shift 2
mkdir -p $O
Q="--no-cpu-baseline --concurrent 0 --steps ${AB_STEPS:-20} --warmup 2 --no-side-metrics --no-strong --log2-cons ${AB_LOG2:-20}"
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}; libenv=""
    # "NAME@DIR:opts": the variant runs with the libraries of spartan_amd/DIR (a variant build of libspartan_hip.so next to a copy of libspartan_host.so)
    if [[ "$name" == *@* ]]; then d=${name#*@}; name=${name%%@*}; libenv="SPARTAN_HIP_LIB=$(pwd)/spartan_amd/$d/libspartan_hip.so SPARTAN_HOST_LIB=$(pwd)/spartan_amd/$d/libspartan_host.so"; fi
    env $libenv SPARTAN_OPTIONS="testing.unlock=1${envs:+,$envs}" BENCH_NO_GATHER_PROBE=1 timeout ${AB_TIMEOUT:-200} python bench.py $Q > $O/ab_${name}_$rep.json 2>$O/ab_${name}_$rep.err
  done
done
python - $O <<'PY'
import json, glob, sys, collections, statistics
d = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/ab_*.json")):
    name = f.split("/ab_")[1].rsplit("_", 1)[0]
    try:
        j = json.load(open(f)); d[name].append((j["ms_per_step"], j["config"]["resident_assignment"]["ms_per_step"], j["config"].get("matches_oracle_digest"), j["config"]["fs_trips_per_proof"], j["phases_ms"]))
    except Exception as e:
        d[name].append((float("nan"), float("nan"), "ERR " + str(e)[:80], 0, {}))
for name, v in d.items():
    ms = [x[0] for x in v]; rs = [x[1] for x in v]
    print("%-24s ms/step (host vars) min %.3f med %.3f | resident min %.3f med %.3f | digest ok %s | trips %s" % (name, min(ms), statistics.median(ms), min(rs), statistics.median(rs), [x[2] for x in v], v[0][3]))
    print("     phases of run 1:", {k: round(x, 2) for k, x in v[0][4].items()})
PY
