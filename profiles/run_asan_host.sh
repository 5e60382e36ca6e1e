#!/bin/bash
# Host side of libtbk under AddressSanitizer + UBSan, in the CPU container (GPU sanitizers are not available on the pool):
#   make -C pythtb_amd/csrc asan  &&  bash profiles/run_asan_host.sh
# runs the CPU test-suite's host tests (symbol table, error paths, knob parsing, the model flatten step on 8
# reference models) against pythtb_amd/csrc/build/asan/libtbk_asan.so.  Any sanitizer report fails the run.
set -eu
cd "$(dirname "$0")/.."
ASANRT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$ASANRT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  TBK_LIBRARY=$PWD/pythtb_amd/csrc/build/asan/libtbk_asan.so python -m pytest tests/test_host_cpu.py -x -q
