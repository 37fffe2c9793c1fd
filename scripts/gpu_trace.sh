#!/bin/bash
# kernel trace of the bench step -> gpurun_out/$1/kernel_stats.txt (rocprofv3 --kernel-trace --stats), extra env in $2..
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/$1; shift
mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --concurrent 0 --steps 8 --warmup 1 --no-side-metrics --no-strong"
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats.log 2>&1
cd $R
largest() { find "$1" -name "$2" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
python profiles/summarize.py "$(largest $O/stats "*_results.db")" --detail k_ipa_round,k_ipa_c0,k_msm_reduce,k_pt_encode,k_msm_rows,k_msm_windows_tree_fused,k_cubic_bind_eval_batched,k_cubic_eval_batched > $O/kernel_stats.txt 2>$O/summarize.err
rm -rf $O/stats
head -30 $O/kernel_stats.txt; sed -n "/per grid/,\$p" $O/kernel_stats.txt; cat $O/summarize.err
