#!/bin/bash
# HBM traffic of a 2^22 proof (BASELINE config 5's size) by kernel: two PMC passes, kernel trace only.
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/p22; mkdir -p $O
B="python $R/bench.py --log2-cons 22 --no-cpu-baseline --concurrent 0 --steps 2 --warmup 1 --no-side-metrics --no-strong"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- $B > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc_$c -name "*counter_collection.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  cp $f $O/pmc_$c.csv; rm -rf $O/pmc_$c
done
ls -la $O
