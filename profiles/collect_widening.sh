#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  bash profiles/collect_widening.sh rNN
# rocprofv3 kernel statistics of the paths outside the headline config: wide-matrix batches (block Jacobi),
# Wilson-loop eigenphases of many bands, per-link LU.  Summaries land in gpurun_out/summary_<tag>/.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
SUM=$REPO/gpurun_out/summary_$TAG
mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/blocked -- python3 $REPO/profiles/blocked_bench.py 64x512 150x101 > $SUM/blocked_bench.txt 2> $OUT/blocked.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wilson -- python3 $REPO/profiles/wilson_big_bench.py > $SUM/wilson_big_bench.txt 2> $OUT/wilson.err
cd $REPO
for w in blocked wilson; do
  f=$(find $OUT/$w -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -25 "$f" > $SUM/${w}_kernel_stats.csv
done
ls -la $SUM
