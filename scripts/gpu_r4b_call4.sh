#!/bin/bash
# eq table as a factor in the throughput-sized batched rounds: kernel test, full parity suite, A/B against the generic kernels
R=$(pwd); O=$R/gpurun_out/r4b4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "eq_table or batched_cubic" > $O/pytest_eq.txt 2>&1; echo "rc $?" >> $O/pytest_eq.txt; tail -15 $O/pytest_eq.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -6 $O/pytest.txt
bash scripts/gpu_ab.sh r4b4 3 "eqf:" "generic:SPARTAN_NO_EQ_FACTOR=1" > $O/ab_eqf.txt 2>&1
cat $O/ab_eqf.txt
python bench.py --log2-cons 22 --no-cpu-baseline --steps 6 --concurrent 0 --no-side-metrics --no-strong > $O/b22_eqf.json 2>$O/b22_eqf.err; python -c "
import json; d=json.load(open('$O/b22_eqf.json')); print('2^22 eqf', d['ms_per_step'], d['config'].get('matches_oracle_digest'))"
SPARTAN_NO_EQ_FACTOR=1 python bench.py --log2-cons 22 --no-cpu-baseline --steps 6 --concurrent 0 --no-side-metrics --no-strong > $O/b22_gen.json 2>$O/b22_gen.err; python -c "
import json; d=json.load(open('$O/b22_gen.json')); print('2^22 generic', d['ms_per_step'], d['config'].get('matches_oracle_digest'))"
