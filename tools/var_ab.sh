#!/bin/bash
# time bench_score.py under each variant library in _lib/variants (env from the caller applies to all)
out=gpurun_out/${1:-varab}; mkdir -p $out
export TMPDIR=/tmp
V=video-retake_amd/retake/_lib/variants
for rep in 1 2; do
for f in $V/libretake_hip_*.so; do
  echo "== $(basename $f)"
  RETAKE_HIP_LIB=$PWD/$f timeout 120 python tools/bench_score.py --iters 30 2>&1 < /dev/null | grep -E "score_pass"
done; done | tee $out/ab.txt
