#!/bin/bash
# same-box A/B of the split-count variants in _lib/variants (tools/variants.sh) at both geometries
out=gpurun_out/${1:-splitab}; mkdir -p $out
export TMPDIR=/tmp
V=video-retake_amd/retake/_lib/variants
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels_timed_region"]
print(f"{d['value']:9.1f} frames/s  p1 {k['score_pass1']['avg_us']:8.1f} us  p2 {k['score_pass2']['avg_us']:8.1f} us")
PY
}
for rep in 1 2; do
for geo in qwen448 baseline; do
  steps=3; [ $geo = baseline ] && steps=1
  for f in default $V/libretake_hip_*.so; do
    n=$(basename $f .so); n=${n#libretake_hip_}
    if [ $f = default ]; then unset RETAKE_HIP_LIB; else export RETAKE_HIP_LIB=$PWD/$f; fi
    timeout 300 python bench.py --geometry $geo --steps $steps --warmup 1 --no-cpu-baseline --no-extras --report $out/$geo.$n.$rep.json > $out/$geo.$n.$rep.line 2> $out/$geo.$n.$rep.err < /dev/null
    echo -n "$geo rep$rep $n: "; summ $out/$geo.$n.$rep.json 2>&1 | tail -1
  done
done; done | tee $out/ab.txt
