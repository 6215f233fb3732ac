#!/bin/bash
# where do the 16-18 us of score_pass1_fixup_kernel go?  (variants: tools/variants.sh fix_base "" fix_p1 "-DRTK_FIXUP_PROBE=1" fix_p2 "-DRTK_FIXUP_PROBE=2")
# the three launches timed as "score_finalize" are the fix-up, the live-key lists and the finalize: the fix-up's share is the difference
for v in ${FIXV:-fix_base fix_p1 fix_p2 fix_base}; do
  for geo in baseline qwen448; do
    RETAKE_HIP_LIB=$PWD/video-retake_amd/retake/_lib/variants/libretake_hip_$v.so python bench.py --geometry $geo --frames 512 --no-extras --no-cpu-baseline --no-self-check --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_untimed_single_stream']['score_finalize']
print('$v $geo: fix-up + key lists + finalize = %.1f us per chunk (%d launches)' % (k['total_ms']*1e3/ (k['launches']/3), k['launches']))"
  done
done
