#!/usr/bin/env python3
"""Per kernel AND launch size: duration (rocprofv3 kernel trace) x HBM bytes (two PMC passes, FETCH_SIZE doubled as the guide's gfx950 note
prescribes + WRITE_SIZE) -> TB/s of every launch class of the F_q streaming kernels, the throughput-sized ones (>= 64 MB moved) apart from
the launch-sized ones. VERDICT r5 #2 asked for exactly this table: bytes / duration per launch size, and what limits each kernel.
With a fifth argument (the SQ pass: SQ_ACTIVE_INST_VALU, SQ_WAIT_INST_ANY, SQ_WAVE_CYCLES — quad-cycles summed over the launch's wavefronts) two more
columns say what a launch below HBM speed spends its time on: valu = share of its wavefronts' cycles spent issuing vector arithmetic (with w
wavefronts per SIMD sharing one vector unit, a share near 1 / w and above means the launch is bound by its field multiplications, not by HBM),
wait = share spent stalled at issue (SQ_WAIT_INST_ANY: the memory pipe / dependencies).
usage: python profiles/fq_bandwidth.py <kernel-trace results.db> <FETCH_SIZE csv> <WRITE_SIZE csv> [resources.txt [SQ csv]]"""
import csv, sqlite3, sys, collections, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
low = {c.lower(): c for c in cols}
def pick(*names):
    for n in names:
        if n in low: return low[n]
    return None
gx, gy, gz = pick("grid_size_x", "grid_x"), pick("grid_size_y", "grid_y"), pick("grid_size_z", "grid_z")
gtot = pick("grid_size")
dcol = low.get("duration") or "(end - start)"
gexpr = gtot if gtot and not gx else "*".join(c for c in (gx, gy, gz) if c)
dur = collections.defaultdict(list)
for name, g, d in db.execute("select name, %s, %s from kernels" % (gexpr, dcol)):
    k = name.split("(")[0].replace("void ", "")
    dur[(k, int(g))].append(d / 1e3)
def load(path, counter):
    tot = collections.defaultdict(float); n = collections.Counter()
    rd = csv.DictReader(open(path))
    gcols = [c for c in rd.fieldnames if c.lower().startswith("grid_size")]
    for row in rd:
        if row["Counter_Name"] != counter: continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        g = 1
        for c in gcols:
            try: g *= max(1, int(float(row[c])))
            except ValueError: pass
        tot[(k, g)] += float(row["Counter_Value"]); n[(k, g)] += 1
    return tot, n
f, nf = load(sys.argv[2], "FETCH_SIZE"); w, nw = load(sys.argv[3], "WRITE_SIZE")
res = {}
if len(sys.argv) > 4:
    for line in open(sys.argv[4]):
        m = re.match(r"(\S+)\s+vgprs\s+(\d+)\s+waves_per_simd\s+(\d+)", line)
        if m: res[m.group(1)] = (int(m.group(2)), int(m.group(3)))
sq = {}
if len(sys.argv) > 5:
    for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"):
        try: sq[c] = load(sys.argv[5], c)
        except OSError: sq = {}; break
def sq_share(k, g, c):
    if not sq: return None
    num, nn = sq[c]; den, nd = sq["SQ_WAVE_CYCLES"]
    if not nn.get((k, g)) or not den.get((k, g)): return None
    return (num[(k, g)] / nn[(k, g)]) / (den[(k, g)] / nd[(k, g)])
FQ = ("k_sc_", "k_cubic_", "k_vecmat", "k_dot", "k_hash_layer", "k_eval_table", "k_sparse_eval", "k_eq_outer", "k_prod_layer", "k_gather", "k_evaluate", "k_spmv", "k_colsum", "k_bind_top", "k_from_index", "k_scale_prefix")
rows = []
for (k, g), ds in dur.items():
    if not k.startswith(FQ): continue
    fb = 2.0 * f.get((k, g), 0.0) * 1024 / max(nf.get((k, g), 1), 1)
    wb = w.get((k, g), 0.0) * 1024 / max(nw.get((k, g), 1), 1)
    avg = sorted(ds)[len(ds) // 2]   # the MEDIAN launch: the first proof of a process pays its pool allocations inside some launches (one 7.5 ms outlier per run)
    rows.append((k, g, len(ds), avg, fb + wb))
print("# F_q kernels by launch size: HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, KiB counters) / median duration (kernel trace)")
print("# throughput-sized = a launch that moves >= 64 MB; the rest is launch-sized (latency-bound by construction: its TB/s says nothing)")
hdr = "%-44s %10s %6s %10s %10s %8s  %-20s %s" % ("kernel", "grid", "calls", "median_us", "MB/launch", "TB/s", "vgprs/waves_per_simd", "valu  wait" if sq else "")
for title, sel in (("throughput-sized launches (>= 64 MB)", lambda r: r[4] >= 64e6), ("launch-sized (< 64 MB), the ten with the most total time", lambda r: r[4] < 64e6)):
    print("\n## " + title); print(hdr)
    part = [r for r in rows if sel(r)]
    part.sort(key=lambda r: -r[2] * r[3])
    if "launch-sized" in title: part = part[:10]
    for k, g, n, avg, b in part:
        base = k.split("<")[0]
        rv = res.get(k) or res.get(base)
        va, wa = sq_share(k, g, "SQ_ACTIVE_INST_VALU"), sq_share(k, g, "SQ_WAIT_INST_ANY")
        tail = ("%.2f  %.2f" % (va, wa)) if (va is not None and wa is not None) else ""
        print("%-44s %10d %6d %10.1f %10.1f %8.2f  %-20s %s" % (k[:44], g, n, avg, b / 1e6, b / avg / 1e6 if avg else 0.0, ("%d / %d" % rv) if rv else "-", tail))
tot_t = sum(r[2] * r[3] for r in rows if r[4] >= 64e6); tot_b = sum(r[2] * r[4] for r in rows if r[4] >= 64e6)
if tot_t: print("\n# all throughput-sized F_q launches together: %.1f ms, %.1f GB -> %.2f TB/s" % (tot_t / 1e3, tot_b / 1e9, tot_b / tot_t / 1e6))
