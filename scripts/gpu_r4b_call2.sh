#!/bin/bash
# host-exclusive time per driver function (diagnostic build) and wall time per C-ABI entry point, one 2^20 proof each
R=$(pwd); O=$R/gpurun_out/r4b2; mkdir -p $O
export TMPDIR=/tmp
Q="--no-cpu-baseline --concurrent 0 --steps 3 --warmup 1 --no-side-metrics --no-strong"
SPARTAN_HOST_LIB=$R/spartan_amd/lib/libspartan_host_prof.so timeout 300 python bench.py $Q > $O/hostprof.json 2> $O/hostprof.err
grep hostprof $O/hostprof.err | tail -45
SPARTAN_CALLSTATS=1 timeout 300 python bench.py $Q > $O/callstats.json 2> $O/callstats.err
grep callstats $O/callstats.err | tail -40
