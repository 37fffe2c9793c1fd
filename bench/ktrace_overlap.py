"""Reads a rocprofv3 kernel-trace CSV and prints, for the kernels of the foreground queue, count / mean / max duration split by
whether a k_msm_rows_bg launch was running at the time. Usage: python bench/ktrace_overlap.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ker = [(r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
bg = [(s, e) for n, s, e in ker if n.startswith("k_msm_rows_bg")]
agg = defaultdict(lambda: [[0, 0.0, 0.0], [0, 0.0, 0.0]])
for n, s, e in ker:
    if n.startswith("k_msm_rows_bg"): continue
    under = any(b0 <= s and e <= b1 for b0, b1 in bg)
    a = agg[n][1 if under else 0]
    d = (e - s) / 1e3
    a[0] += 1; a[1] += d; a[2] = max(a[2], d)
print("%-46s %22s   %22s" % ("kernel", "alone: n / mean / max us", "under bg: n / mean / max us"))
for n, (a, b) in sorted(agg.items(), key=lambda kv: -kv[1][1][1]):
    if b[0]:
        print("%-46s %6d %7.1f %7.1f   %6d %7.1f %7.1f" % (n, a[0], a[1] / max(a[0], 1), a[2], b[0], b[1] / b[0], b[2]))
