out=gpurun_out/r12c; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -s -k "reference" 2>&1 | grep -E "passed|failed|differ|Error|assert" | head -30
B="--score-rounding reference --no-extras --no-cpu-baseline --steps 2 --warmup 1 --no-self-check"
for v in prod p1nb1 p2nb2; do
  if [ $v = prod ]; then unset RETAKE_HIP_LIB; else export RETAKE_HIP_LIB=$PWD/video-retake_amd/retake/_lib/variants/libretake_hip_$v.so; fi
  timeout 600 python bench.py $B > $out/ref_$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$out/ref_$v.json").read().strip().splitlines()[-1])
k=d["kernels_timed_region"]
print("$v", round(d["value"],1), "frames/s  pass1", round(k["score_pass1"]["avg_us"]), "pass2", round(k["score_pass2"]["avg_us"]), "frac", round(d["roofline"]["frac"],3))
PY
done
