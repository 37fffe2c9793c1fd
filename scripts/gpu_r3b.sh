#!/bin/bash
# round-3 GPU batch B: probes (FP64-FMA multiplication, device Keccak), kernel trace of the bench step, window-width A/B
set -u
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r3b
mkdir -p $O
timeout 120 bench/ubench_fpmul > $O/ubench_fpmul.txt 2>&1
timeout 60 bench/keccak_probe > $O/keccak_probe.txt 2>&1
B="python $R/bench.py --no-cpu-baseline --concurrent 0 --steps 4 --warmup 1 --no-side-metrics --no-strong"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats.log 2>&1
cd $R
largest() { find "$1" -name "$2" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
python profiles/summarize.py "$(largest $O/stats '*_results.db')" > $O/kernel_stats.txt 2>$O/summarize.err
python - "$(largest $O/stats '*_results.db')" > $O/ipa_round_durations.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print(tabs)
try:
    rows = list(db.execute("select name, start, end from kernels order by start"))
    import collections
    d = collections.defaultdict(list)
    for n, s, e in rows:
        d[n.split('(')[0][:50]].append((e - s) / 1e3)
    for k, v in d.items():
        if 'ipa' in k or 'msm_windows' in k or 'msm_reduce' in k:
            v2 = sorted(v)
            print(k, len(v), "min %.1f med %.1f max %.1f us" % (v2[0], v2[len(v2)//2], v2[-1]))
except Exception as ex:
    print("ERR", ex)
PY
rm -rf $O/stats
Q="--no-cpu-baseline --concurrent 0 --steps 20 --warmup 2 --no-side-metrics --no-strong"
for rep in 1 2 3; do
  SPARTAN_MSM_WIDE_GB=200 timeout 200 python bench.py $Q > $O/w_15_15_$rep.json 2>/dev/null
  timeout 200 python bench.py $Q > $O/w_15_14_$rep.json 2>/dev/null
  SPARTAN_MSM_WBITS=14 timeout 200 python bench.py $Q > $O/w_14_14_$rep.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3b/w_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"], 3), d["config"]["resident_assignment"]["ms_per_step"], d["config"]["table_GB"]["window_bits"], d["phases_ms"]["commit_nondet_witness"], d["phases_ms"]["polycommit"])
    except Exception as e: print(f, "ERR", e)
PY
cat $O/ubench_fpmul.txt | head -30; cat $O/keccak_probe.txt; head -40 $O/kernel_stats.txt; cat $O/ipa_round_durations.txt
