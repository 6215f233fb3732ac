#!/bin/bash
# FAST mode: parity tests, then same-box timing of exact vs fast (and the variant libraries' fast) at both geometries
out=gpurun_out/${1:-fastab}; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -s -k "fast or bf16_against_reference" > $out/pytest_fast.log 2>&1
grep -E "passed|failed|fast rounding|mode vs the reference" $out/pytest_fast.log | tail -20
V=video-retake_amd/retake/_lib/variants
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels_timed_region"]
print(f"{d['value']:9.1f} frames/s  p1 {k['score_pass1']['avg_us']:8.1f}  p2 {k['score_pass2']['avg_us']:8.1f}")
PY
}
for rep in 1 2; do
for geo in baseline qwen448; do
  steps=3; [ $geo = baseline ] && steps=1
  for mode in fp32 fast; do
  for f in default $V/libretake_hip_*.so; do
    n=$(basename $f .so); n=${n#libretake_hip_}
    [ $mode = fp32 ] && [ $f != default ] && continue
    if [ $f = default ]; then unset RETAKE_HIP_LIB; else export RETAKE_HIP_LIB=$PWD/$f; fi
    timeout 300 python bench.py --geometry $geo --score-rounding $mode --steps $steps --warmup 1 --no-cpu-baseline --no-extras --no-self-check --report $out/$geo.$mode.$n.$rep.json > $out/$geo.$mode.$n.$rep.line 2> $out/$geo.$mode.$n.$rep.err < /dev/null
    echo -n "$geo rep$rep $mode $n: "; summ $out/$geo.$mode.$n.$rep.json 2>&1 | tail -1
  done; done
done; done | tee $out/ab.txt
