#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4b5; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_large.py tests/test_golden.py tests/test_gpu_proofs.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
bash scripts/gpu_ab.sh r4b5 3 "eqf:" "generic:SPARTAN_NO_EQ_FACTOR=1" > $O/ab_eqf.txt 2>&1
cat $O/ab_eqf.txt
B="python $R/bench.py --no-cpu-baseline --concurrent 0 --steps 4 --warmup 1 --no-side-metrics --no-strong"
export BENCH_NO_GATHER_PROBE=1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats.log 2>&1
cd $R
largest() { find "$1" -name "$2" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
python profiles/summarize.py "$(largest $O/stats '*_results.db')" --detail k_cubic_bind_eval_batched,k_cubic_eval_batched > $O/kernel_stats.txt 2>$O/summarize.err
rm -rf $O/stats
head -40 $O/kernel_stats.txt
