#!/bin/bash
# VGPRs / scratch / occupancy of every kernel instantiation in the library, from the compiler's own remarks
# (-Rpass-analysis=kernel-resource-usage; needs no GPU).   usage: tools/resource_usage.sh > profiles/rNN_kernel_resources.txt
R=$(cd "$(dirname "$0")/.." && pwd)
echo "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage, every __global__ instantiation of krep_amd/csrc/*.hip"
echo "# file | kernel | VGPRs | scratch bytes/lane | waves/SIMD | LDS bytes (static)"
for f in "$R"/krep_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-cuda-compat -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - |
    sed 's/Function Name: //' | while IFS=$'\t' read -r name v s o l; do
      printf "%s | %s | %s | %s | %s | %s\n" "$(basename "$f")" "$(echo "$name" | c++filt | sed 's/(.*//; s/^void //; s/kg:://g')" \
        "${v#*: }" "${s#*: }" "${o#*: }" "${l#*: }"
    done
done
