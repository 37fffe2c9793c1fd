#!/bin/bash
# VGPRs and occupancy of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed) -> profiles/r6_kernel_resources.txt
cd "$(dirname "$0")/../spartan_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Occupancy" | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - | \
    awk '{print $3, $5, $8}' | while read name v o; do echo "$(echo $name | c++filt | sed 's/(.*//; s/^void //') vgprs $v waves_per_simd $o"; done
done | sort -u
rm -f /tmp/kr_$$.o
