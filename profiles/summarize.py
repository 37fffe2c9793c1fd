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
