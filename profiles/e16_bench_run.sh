#!/bin/bash
# Run on the GPU box from the repo root: quality legs (against the three-kernel path and a host Jacobi reference) and the mesh
# timing of profiles/microbench/e16_bench.   bash profiles/e16_bench_run.sh [binary]
BIN=${1:-profiles/microbench/e16_bench}
for kind in 0 1 2 3; do timeout 120 $BIN 137312 16 $kind 3 | grep -v "^$"; done
timeout 120 $BIN 20000 11 0 3 | grep -v "^$"
timeout 120 $BIN mesh 65 5
