#!/usr/bin/env python3
"""Per-kernel HBM traffic from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B
requests at 64 B, i.e. reports half the bytes of a wide (16 B/lane) coalesced stream -> doubled. All kernels here load
16 B per lane (ulonglong2). WRITE_SIZE is uncalibrated on gfx950 (reported as is).
usage: python profiles/pmc_summarize.py <FETCH_SIZE pass>.csv <WRITE_SIZE pass>.csv <name of the committed text summary> <log2_cons> <SPARTAN_OPTIONS of the passes>
Adds one entry to profiles/pmc_traffic.json, keyed by the configuration it was collected for (kernel sources | size | non-default
options: bench.config_key): bench.py reports `roofline.traffic` only for a run whose own key has an entry.
(An option given to the passes must not equal its default: the bench builds its key from the options that differ from theirs.)"""
import csv, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FAMILY = {"k_msm_rows": "msm_rows_fixed", "k_msm_rows_bg": "msm_rows_fixed", "k_msm_flat": "msm_rows_fixed", "k_msm_flat_bg": "msm_rows_fixed", "k_msm_lds": "msm_rows_fixed", "k_msm_q": "msm_rows_fixed", "k_msm_windows": "msm_windows_fixed", "k_msm_windows_tree": "msm_windows_fixed", "k_msm_windows_tree_fused": "msm_windows_fixed", "k_ipa_round": "ipa_round",
          "k_msm_reduce": "msm_reduce_compress", "k_pt_encode": "msm_reduce_compress", "k_pt_encode_lean": "msm_reduce_compress", "k_pt_reduce_pass": "msm_reduce_pass",
          "k_cubic_bind_eval_batched": "sumcheck_bind_eval", "k_sc_bind_eval": "sumcheck_bind_eval", "k_sc_eval": "sumcheck_eval",
          "k_cubic_eval_batched": "sumcheck_eval", "k_cubic_bind_eval_batched_eq": "sumcheck_bind_eval", "k_cubic_eval_batched_eq": "sumcheck_eval", "k_bind_top": "table_bind", "k_eq_expand": "eq_expand",
          "k_cubic_bind2_eval": "sumcheck_bind_eval", "k_cubic_bind_eval_tiny": "sumcheck_bind_eval", "k_sc_bind_eval_tiny": "sumcheck_bind_eval",
          "k_cubic_eval_tiny": "sumcheck_eval", "k_vecmat": "vecmat", "k_dot_many": "dot", "k_dot3": "dot", "k_dot": "dot"}
def load(path, counter):
    tot = collections.defaultdict(float); n = collections.Counter()
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter: continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        tot[k] += float(row["Counter_Value"]); n[k] += 1
    return tot, n
f, nf = load(sys.argv[1], "FETCH_SIZE"); w, nw = load(sys.argv[2], "WRITE_SIZE")
print("%-32s %8s %16s %16s %16s" % ("kernel", "launches", "fetch_B/launch", "write_B/launch", "hbm_B/launch"))
fam_bytes = collections.defaultdict(float); fam_n = collections.Counter()
for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0))):
    fb = 2.0 * f[k] * 1024 / nf[k]; wb = w.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
    print("%-32s %8d %16.0f %16.0f %16.0f" % (k[:32], nf[k], fb, wb, fb + wb))
    if k in FAMILY:
        fam_bytes[FAMILY[k]] += 2.0 * f[k] * 1024 + w.get(k, 0.0) * 1024
        fam_n[FAMILY[k]] += nf[k]
out = {fam: fam_bytes[fam] / max(fam_n[fam], 1) for fam in fam_bytes}
# msm_rows_fixed per launch SHAPE is what the bench line's dominant-kernel entry averages over: keep the two kernels apart too
for k in ("k_msm_rows", "k_msm_rows_bg", "k_msm_flat", "k_msm_flat_bg", "k_msm_lds", "k_msm_q"):
    if k in f:
        out[k] = (2.0 * f[k] * 1024 + w.get(k, 0.0) * 1024) / nf[k]
from bench import kernel_source_digest
# keyed by configuration (round 6): kernel sources | instance size | the non-default library options of the run — the same
# string bench.config_key builds from its context. argv[4] = log2_cons, argv[5] = the SPARTAN_OPTIONS the passes ran with ("" = defaults)
log2 = int(sys.argv[4]) if len(sys.argv) > 4 else 20
opts = sorted(o for o in (sys.argv[5] if len(sys.argv) > 5 else "").split(",") if o and not o.startswith(("testing.unlock", "host.callstats", "debug.ktime")))
key = "%s|2^%d|%s" % (kernel_source_digest(), log2, ",".join(opts) or "defaults")
out["source"] = sys.argv[3] if len(sys.argv) > 3 else "pmc_traffic.json"
path = "profiles/pmc_traffic.json"
try:
    allj = json.load(open(path))
    if "entries" not in allj: allj = {"entries": {}}
except (OSError, ValueError):
    allj = {"entries": {}}
allj["entries"][key] = out
json.dump(allj, open(path, "w"), indent=1)
print("\nper-family HBM bytes per launch (profiles/pmc_traffic.json, entry %s):" % key, json.dumps(out))
