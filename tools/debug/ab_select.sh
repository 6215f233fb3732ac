#!/bin/bash
# same-box A/B: batched selection kernel, register-resident (RTK_SEL_LDS=0) vs the small-code LDS kernel
for v in sel_reg sel_lds sel_reg sel_lds; do
  lib=video-retake_amd/retake/_lib/variants/libretake_hip_$v.so
  for geo in baseline qwen448; do
    RETAKE_HIP_LIB=$PWD/$lib python bench.py --geometry $geo --frames 512 --no-extras --no-cpu-baseline --no-self-check --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_untimed_single_stream']
print('$v $geo: select %.1f us  compact %.1f us  finalize %.1f us   %.1f frames/s' % (k['pivotkv_select']['avg_us'], k['compact_units']['avg_us'], k['score_finalize']['avg_us'], d['value']))"
  done
done
