#!/bin/bash
# Same-box A/B of the per-update kernels' launch shape on top of the buffer addressing (variants.h: RTK_PREP_NW / _BLOCK / _YSPLIT):
#   tools/variants.sh nw2 "-DRTK_PREP_NW=2" b128 "-DRTK_PREP_BLOCK=128" ...; tools/prep_shape_ab.sh nw2 b128 ...
out=gpurun_out/prep_shape_ab; mkdir -p $out
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kernels_untimed_single_stream"]
name = "prologue" if "prologue" in k else "unrotate_pack"
print(f"{sys.argv[2]:40s} {d['value']:9.1f} frames/s  {d['ms_per_step']:8.2f} ms/step  {name} {k[name]['avg_us']:7.2f} us")
PY
}
for rep in 1 2; do
  for tag in intree "$@"; do
    lib=$([ $tag = intree ] && echo "" || echo video-retake_amd/retake/_lib/variants/libretake_hip_$tag.so)
    for mode in update prerope; do
      extra=$([ $mode = prerope ] && echo --pre-rope)
      for geo in qwen448 baseline; do
        [ $geo = baseline ] && [ $rep != 1 ] && continue
        RETAKE_HIP_LIB=$lib timeout 300 python bench.py --geometry $geo --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-self-check $extra \
            --report $out/$tag.$mode.$geo.$rep.json > /dev/null 2> $out/$tag.$mode.$geo.$rep.err < /dev/null
        summ $out/$tag.$mode.$geo.$rep.json "$tag $mode $geo rep$rep" | tee -a $out/summary.txt
      done
    done
  done
done
