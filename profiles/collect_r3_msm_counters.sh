#!/bin/bash
# address-translation and fabric counters of k_msm_rows on the 4098-point table set at 13 / 14 / 15-bit windows
# (40 / 61 / 110 GB of tables). One counter group per rocprofv3 pass, --kernel-trace only. Output: gpurun_out/r3msm/*.csv + summary.
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r3msm; mkdir -p $O
cd /tmp
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
grep -o "TCP_UTCL1[A-Z_0-9a-z]*\|TCC_EA0_RDREQ[A-Za-z_0-9]*\|TCC_TAG_STALL[A-Za-z_0-9]*\|UTCL2[A-Za-z_0-9]*\|TCP_PENDING[A-Za-z_0-9]*\|TCC_BUSY[A-Za-z_0-9]*\|TCP_TCC_READ_REQ[A-Za-z_0-9]*" $O/list_avail.txt | sort -u > $O/candidate_counters.txt
largest() { find "$1" -name "$2" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
for bits in 13 14 15; do
  n=0
  for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" "FETCH_SIZE"; do
    n=$((n+1))
    SPARTAN_MSM_WBITS=$bits timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p_${bits}_$n -- python $R/bench/msm_probe.py > $O/p_${bits}_$n.log 2>&1
    f=$(largest $O/p_${bits}_$n '*counter_collection.csv'); [ -n "$f" ] && cp "$f" $O/pmc_${bits}_$n.csv
    rm -rf $O/p_${bits}_$n
  done
done
cd $R
for bits in 13 14 15; do echo "== $bits-bit windows"; python profiles/pmc_counters.py $O/pmc_${bits}_*.csv --kernels k_msm_rows; grep commit_rows $O/p_${bits}_1.log | tail -1; done > $O/summary.txt 2>&1
cat $O/summary.txt; head -30 $O/candidate_counters.txt
