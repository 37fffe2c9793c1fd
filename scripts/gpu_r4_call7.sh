#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4c7; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_golden.py tests/test_gpu_proofs.py tests/test_gpu_properties.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -5 $O/pytest.txt
python - <<'PY' > $O/encode.txt 2>&1
import time, sys
sys.path.insert(0, ".")
from spartan_amd import prover as P
ctx = P.Ctx(0)
for s in (20, 22):
    N = 1 << s
    inst = P.Instance.produce_synthetic_r1cs(ctx, N, N, 10, seed=0)
    gens = P.SNARKGens(ctx, N, N, 10, N)
    for k in range(4):
        t0 = time.perf_counter(); e = P.SNARK.encode(ctx, inst, gens); dt = time.perf_counter() - t0; e.free()
        print("SNARK::encode 2^%d: %.2f ms" % (s, dt * 1e3))
    gens.free(); inst.free()
PY
cat $O/encode.txt
