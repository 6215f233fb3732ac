#!/bin/bash
# timing ablations of score_pass2_occ_kernel (variants built by tools/variants.sh)
out=gpurun_out/${1:-occab}; mkdir -p $out
export TMPDIR=/tmp
V=video-retake_amd/retake/_lib/variants
for name in base nobar nostage nols nofrag nosm nomfma nobar_nostage nofrag_nols all_lds_off; do
  f=$V/libretake_hip_$name.so
  [ -f $f ] || continue
  echo "== $name"
  RETAKE_HIP_LIB=$PWD/$f RTK_PASS2_OCC=1 timeout 120 python tools/bench_score.py --iters 20 2>&1 < /dev/null | grep -E "score_pass2"
done | tee $out/ablate.txt
