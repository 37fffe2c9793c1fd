#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel-trace database into the text summary committed under profiles/.
usage: python profiles/summarize.py gpurun_out/prof_xx/<name>_results.db > profiles/<name>_kernel_stats.txt"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)")
print("%-60s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
import collections
agg = collections.OrderedDict()
for name, calls, total, avg, pct in rows:
    short = name.split("(")[0].replace("void ", "").split("<")[0]  # template instantiations of one kernel are merged
    c, t, p = agg.get(short, (0, 0.0, 0.0))
    agg[short] = (c + calls, t + total, p + pct)
for short, (calls, total, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %8d %14.1f %12.3f %6.2f%%" % (short[:60], calls, total, total / calls, pct))
# optional: --detail k_a,k_b  -> per-grid-size statistics of those kernels (which launches of a family are the slow ones)
if "--detail" in sys.argv:
    want = sys.argv[sys.argv.index("--detail") + 1].split(",")
    cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
    gcol = [c for c in cols if c.lower() in ("grid_size_x", "grid_x", "grid_size")]
    dcol = "duration" if "duration" in cols else "(end - start)"
    if not gcol:
        print("# --detail: no grid column in view `kernels` (%s)" % ",".join(cols))
    else:
        print("\n# per grid size (x): kernel, grid_x, launches, avg_us, min_us, max_us")
        for k in want:
            q = "select %s, count(*), avg(%s), min(%s), max(%s) from kernels where name like ? group by 1 order by 1" % (gcol[0], dcol, dcol, dcol)
            for g, n, a, mn, mx in db.execute(q, ("%" + k + "%",)):
                print("%-36s %10d %6d %10.1f %10.1f %10.1f" % (k, g, n, a / 1e3, mn / 1e3, mx / 1e3))
