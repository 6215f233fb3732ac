#!/bin/bash
# Same-box A/B of the per-update kernels' row addressing (variants.h RTK_PREP_UBASE): the in-tree library (1: buffer
# descriptors, head offset in an SGPR) against libretake_hip_ubase0.so (0: 64-bit row addresses in the vector ALU).
#   tools/variants.sh ubase0 "-DRTK_PREP_UBASE=0"; tools/ubase_ab.sh  ->  gpurun_out/ubase_ab/summary.txt
out=gpurun_out/ubase_ab; mkdir -p $out
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kernels_untimed_single_stream"]
name = "prologue" if "prologue" in k else "unrotate_pack"
print(f"{sys.argv[2]:44s} {d['value']:9.1f} frames/s  {d['ms_per_step']:8.2f} ms/step  {name} {k[name]['avg_us']:7.2f} us")
PY
}
for rep in 1 2 3; do
  for lib in "" video-retake_amd/retake/_lib/variants/libretake_hip_ubase0.so; do
    tag=$([ -z "$lib" ] && echo ubase1 || echo ubase0)
    for mode in update prerope; do
      extra=$([ $mode = prerope ] && echo --pre-rope)
      for geo in qwen448 baseline; do
        [ $geo = baseline ] && [ $rep != 1 ] && continue
        RETAKE_HIP_LIB=$lib timeout 300 python bench.py --geometry $geo --steps 3 --warmup 1 --no-cpu-baseline --no-extras $extra \
            --report $out/$tag.$mode.$geo.$rep.json > /dev/null 2> $out/$tag.$mode.$geo.$rep.err < /dev/null
        summ $out/$tag.$mode.$geo.$rep.json "$tag $mode $geo rep$rep" | tee -a $out/summary.txt
      done
    done
  done
done
