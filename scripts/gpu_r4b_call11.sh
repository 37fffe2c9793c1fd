#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4b11; mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r4b11 3 "default:" "collow:SPARTAN_OVERLAP_COL=1,SPARTAN_COL_LOWPRIO=1" "collow_bg4:SPARTAN_OVERLAP_COL=1,SPARTAN_COL_LOWPRIO=1,SPARTAN_BG_EIGHTHS=4" "colbg:SPARTAN_OVERLAP_COL=1" > $O/ab_col.txt 2>&1
cat $O/ab_col.txt
