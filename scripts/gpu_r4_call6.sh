#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4c6; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest_kernels.txt 2>&1; echo "rc $?" >> $O/pytest_kernels.txt; tail -15 $O/pytest_kernels.txt
python - <<'PY' > $O/encode.txt 2>&1
import time, sys
sys.path.insert(0, ".")
from spartan_amd import prover as P
ctx = P.Ctx(0); N = 1 << 20
inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=0)
gens = P.SNARKGens(ctx, N, N, 10, N)
for k in range(4):
    t0 = time.perf_counter(); e = P.SNARK.encode(ctx, inst, gens); dt = time.perf_counter() - t0; e.free()
    print("SNARK::encode 2^20: %.2f ms" % (dt * 1e3))
PY
cat $O/encode.txt
