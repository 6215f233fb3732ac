#!/bin/bash
# same-box A/B of the in-place compaction kernel's shape knobs (variants built by tools/variants.sh beforehand)
for v in base hu2 w5 w3 notab; do
  lib=video-retake_amd/retake/_lib/variants/libretake_hip_cmp_$v.so
  [ -f $lib ] || continue
  for geo in baseline qwen448; do
    RETAKE_HIP_LIB=$PWD/$lib python bench.py --geometry $geo --frames 512 --no-extras --no-cpu-baseline --no-self-check --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_untimed_single_stream']
print('$v $geo: compact %.1f us  select %.1f us   %.1f frames/s' % (k['compact_units']['avg_us'], k['pivotkv_select']['avg_us'], d['value']))"
  done
done
