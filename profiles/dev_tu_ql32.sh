#!/bin/bash
# Development translation unit for k_ql32_lanes: resource usage of <1, 2, 24>, <1, 2, 32>, <0, 0, 32> in seconds.
cd /root/repo/pythtb_amd/csrc
{
  echo '#include <math.h>'; echo '#include <stdlib.h>'; echo '#include <string.h>'; echo '#include <algorithm>'; echo '#include <type_traits>'
  echo '#include "tbk_internal.h"'; echo '#include "tbk_solve_dev.h"'
  sed -n '/^struct QlwWork {/,/^};/p' tbk_solve_qlw.inl
  echo '#include "tbk_solve_ql32.inl"'
  for a in "1, 28" "1, 24" "1, 32" "0, 32"; do
    echo "template __global__ void k_ql32_lanes<$a>(const int, const int64_t, const int64_t, const int64_t, const QlwWork, double*, const GridArgs, int*);"
  done
} > ql32_dev.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value --cuda-device-only -S ql32_dev.hip -o /tmp/ql32_dev.s -Rpass-analysis=kernel-resource-usage "$@" 2> /tmp/ql32_res.txt
grep -o "Function Name: [A-Za-z0-9_]*\|VGPRs: [0-9]*\|AGPRs: [0-9]*\|ScratchSize \[bytes/lane\]: [0-9]*\|Occupancy \[waves/SIMD\]: [0-9]*" /tmp/ql32_res.txt | tr '\n' ' ' | sed 's/Function Name/\nFunction Name/g'; echo
grep error -A4 /tmp/ql32_res.txt | head -20
rm -f ql32_dev.hip
