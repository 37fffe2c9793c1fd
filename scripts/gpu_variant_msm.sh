#!/bin/bash
# row-MSM timing of an experiment variant of the library against the default one: bash scripts/gpu_variant_msm.sh OUTDIR NAME
R=$(pwd); O=$R/gpurun_out/$1; mkdir -p $O; : > $O/msm_variant.txt
for rep in 1 2 3; do
for b in 14 15; do
  for lib in libspartan_hip.so libspartan_hip_$2.so; do
    echo "== $lib wbits $b" >> $O/msm_variant.txt
    SPARTAN_HIP_LIB=$R/spartan_amd/lib/$lib SPARTAN_OPTIONS=msm.wbits=$b timeout 300 python bench/msm_probe.py 2>&1 | tail -2 >> $O/msm_variant.txt
  done
done
done
python - "$O/msm_variant.txt" <<'PY'
import sys, re, collections
best = collections.defaultdict(lambda: 1e9); key = None
for line in open(sys.argv[1]):
    if line.startswith("=="): key = line.strip()
    m = re.search(r"([0-9.]+) ms", line)
    if m and key: best[key] = min(best[key], float(m.group(1)))
for k in sorted(best): print(k, "min %.3f ms" % best[k])
PY
