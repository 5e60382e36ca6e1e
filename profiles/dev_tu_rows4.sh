#!/bin/bash
# Development translation unit: tbk_solve.hip up to the end of k_grid_rows + ONE explicit instantiation (2 s instead of 2 min 20 s);
# extra arguments = compiler flags, e.g. "-DTBK_ROWS_OCC=__attribute__((amdgpu_waves_per_eu(4,8)))" to probe the distance to an
# occupancy step by the spill bytes it forces.  Writes /tmp/rows4_dev.s and prints the kernel's resource usage.
cd /root/repo/pythtb_amd/csrc
END=$(grep -n "^__global__ __launch_bounds__(256) void k_mesh_evals" tbk_solve.hip | cut -d: -f1)
END=$((END-8))
head -n $END tbk_solve.hip | sed '/^#include "tbk_solve_fused.inl"/d' > rows4_dev.hip
echo 'template __global__ void k_grid_rows<4, 1>(const ModelView, const GridArgs);' >> rows4_dev.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -mllvm -disable-machine-licm --cuda-device-only -S rows4_dev.hip -o /tmp/rows4_dev.s -Rpass-analysis=kernel-resource-usage "$@" 2> /tmp/rows4_res.txt
grep -A9 "Function Name: _Z11k_grid_rowsILi4ELi1E" /tmp/rows4_res.txt | grep -o "VGPRs: [0-9]*\|ScratchSize \[bytes/lane\]: [0-9]*\|Occupancy \[waves/SIMD\]: [0-9]*\|VGPRs Spill: [0-9]*\|SGPRs: [0-9]*" | tr '\n' ' '; echo
grep error -A4 /tmp/rows4_res.txt | head -20
rm -f rows4_dev.hip
