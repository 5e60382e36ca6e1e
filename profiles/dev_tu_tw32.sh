#!/bin/bash
# Development translation unit for k_tw32_vectors (seconds instead of the 4 min 40 s of tbk_solve.hip): prints resource usage of <1, 24> and <1, 32>.
cd /root/repo/pythtb_amd/csrc
{
  echo '#include <math.h>'; echo '#include <stdlib.h>'; echo '#include <string.h>'; echo '#include <algorithm>'; echo '#include <type_traits>'
  echo '#include "tbk_internal.h"'; echo '#include "tbk_solve_dev.h"'
  echo 'typedef double tw_d4 __attribute__((ext_vector_type(4)));'
  echo '#define TW_LDS_ORDER() asm volatile("" ::: "memory")'
  sed -n '/^__device__ __forceinline__ double tw_rcp/,/^__device__ __forceinline__ double tw_guard/p' tbk_solve_tw16.inl
  sed -n '/^struct QlwWork {/,/^};/p' tbk_solve_qlw.inl
  grep "^__host__ __device__ constexpr int hh32_rec" tbk_solve_hh32.inl
  sed -n '/^__device__ __forceinline__ double hh32_xhalf/,/^}/p' tbk_solve_hh32.inl
  echo '#include "tbk_solve_tw32.inl"'
  echo 'template __global__ void k_tw32_vectors<1, 28, true>(const int, const int64_t, const int64_t, const int64_t, const QlwWork, cd*, const WfsView);'
  echo 'template __global__ void k_tw32_vectors<1, 32, false>(const int, const int64_t, const int64_t, const int64_t, const QlwWork, cd*, const WfsView);'
  echo 'template __global__ void k_tw32_vectors<1, 32, true>(const int, const int64_t, const int64_t, const int64_t, const QlwWork, cd*, const WfsView);'
} > tw32_dev.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value --cuda-device-only -S tw32_dev.hip -o /tmp/tw32_dev.s -Rpass-analysis=kernel-resource-usage "$@" 2> /tmp/tw32_res.txt
grep -o "Function Name: [A-Za-z0-9_]*\|VGPRs: [0-9]*\|AGPRs: [0-9]*\|ScratchSize \[bytes/lane\]: [0-9]*\|Occupancy \[waves/SIMD\]: [0-9]*" /tmp/tw32_res.txt | tr '\n' ' ' | sed 's/Function Name/\nFunction Name/g'; echo
grep error -A4 /tmp/tw32_res.txt | head -20
rm -f tw32_dev.hip
