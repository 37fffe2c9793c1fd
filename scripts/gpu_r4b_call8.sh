#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r4b8; mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r4b8 2 "bg5:" "bg4:SPARTAN_BG_EIGHTHS=4" "bg6:SPARTAN_BG_EIGHTHS=6" "flatbg5:SPARTAN_MSM_FLAT_BG=1" "flatbg4:SPARTAN_MSM_FLAT_BG=1,SPARTAN_BG_EIGHTHS=4" "dmax2048:SPARTAN_DOUBLE_ROUND_MAX_LEN=2048" "dmax8192:SPARTAN_DOUBLE_ROUND_MAX_LEN=8192" > $O/ab_bg.txt 2>&1
cat $O/ab_bg.txt
