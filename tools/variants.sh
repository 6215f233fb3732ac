#!/bin/bash
# Build A/B variants of libretake_hip.so:  tools/variants.sh name "extra hipcc flags" [name2 "flags2" ...]
# Output: video-retake_amd/retake/_lib/variants/libretake_hip_<name>.so   (select with RETAKE_HIP_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${RTK_SRC:-$ROOT/video-retake_amd/csrc}
OUT=$ROOT/video-retake_amd/retake/_lib/variants
mkdir -p $OUT
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -ffp-contract=on -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form=1"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  tmp=$(mktemp -d)
  for f in api dpselect mallm_chain rope pivotkv_score pivotkv_evict pivotkv_compact pivotkv_update p2p; do
    /opt/rocm/bin/hipcc $BASE $flags -c $SRC/$f.hip -o $tmp/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libretake_hip_$name.so $tmp/*.o
  rm -rf $tmp
  echo built $name
done
