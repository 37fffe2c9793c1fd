#!/bin/bash
# round 4, second session, call 1: parity suite, A/B of the host-side changes (Keccak form, challenge inversion), proofs in flight with the gate
R=$(pwd); O=$R/gpurun_out/r4b1; mkdir -p $O
export TMPDIR=/tmp
grep -m1 "model name" /proc/cpuinfo > $O/host.txt; nproc >> $O/host.txt
python -c "
import ctypes
from spartan_amd import prover
prover.H.spz_keccak_variant.restype = ctypes.c_char_p
print('keccak variant picked:', prover.H.spz_keccak_variant().decode())" >> $O/host.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
bash scripts/gpu_ab.sh r4b1 2 "new:" "oldhost:SPARTAN_KECCAK=plain,SPARTAN_INVERT_CHAIN=1" "bmi2:SPARTAN_KECCAK=bmi2" "avx512:SPARTAN_KECCAK=avx512" > $O/ab_host.txt 2>&1
cat $O/ab_host.txt
timeout 600 python bench/concurrent_probe.py 20 4 8 > $O/concurrent.txt 2>&1
cat $O/concurrent.txt
cat $O/host.txt
