#!/bin/bash
# Window-width A/B at 2^20 on one GPU box (run through gpurun): SNARK::prove with the generator tables forced to c-bit windows.
# c = 6: 43 additions per scalar, tables 132 KB per generator (the 1025-point stream's 135 MB fit the 256 MB Infinity Cache);
# c = 15: 17 additions, 26.7 MB per generator (137 GB for both streams, HBM only).
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/width; mkdir -p $O
for c in 6 8 10 12 13 14 15; do
  SPARTAN_MSM_WBITS=$c python bench.py --no-cpu-baseline --concurrent 0 --steps 10 --no-side-metrics --no-strong 2>/dev/null | tail -1 > $O/line_$c.json
done
cd /tmp
for c in 6 15; do
  for k in FETCH_SIZE WRITE_SIZE; do
    SPARTAN_MSM_WBITS=$c rocprofv3 --kernel-trace --pmc $k --output-format csv -d $O/pmc_${c}_$k -- python $R/bench.py --no-cpu-baseline --concurrent 0 --steps 2 --warmup 1 --no-side-metrics --no-strong > /dev/null 2>&1
    f=$(find $O/pmc_${c}_$k -name "*counter_collection.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
    grep -E "k_msm_rows|Kernel_Name" $f > $O/pmc_${c}_$k.csv; rm -rf $O/pmc_${c}_$k
  done
done
ls -la $O
