#!/usr/bin/env python3
"""Per-kernel means of arbitrary rocprofv3 --pmc counters (one or more pmc_counter_collection CSVs, one pass each).
usage: python profiles/pmc_counters.py <csv> [<csv> ...] [--kernels k_sc_eval,k_vecmat,...]"""
import csv, sys, collections
files = [a for a in sys.argv[1:] if not a.startswith("--")]
want = None
for a in sys.argv[1:]:
    if a.startswith("--kernels"):
        want = set(sys.argv[sys.argv.index(a) + 1].split(",")) if a == "--kernels" else set(a.split("=", 1)[1].split(","))
files = [f for f in files if want is None or f not in (",".join(sorted(want)),)]
val = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter()); counters = []
for path in files:
    try:
        rd = csv.DictReader(open(path))
    except OSError:
        continue
    for row in rd:
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        c = row["Counter_Name"]
        if c not in counters: counters.append(c)
        val[k][c] += float(row["Counter_Value"]); cnt[k][c] += 1
print("%-40s %8s " % ("kernel", "launches") + " ".join("%22s" % c[:22] for c in counters))
for k in sorted(val, key=lambda k: -max(cnt[k].values())):
    base = k.split("<")[0]
    if want is not None and base not in want: continue
    print("%-40s %8d " % (k[:40], max(cnt[k].values())) + " ".join("%22.4g" % (val[k][c] / cnt[k][c]) if cnt[k][c] else "%22s" % "-" for c in counters))
