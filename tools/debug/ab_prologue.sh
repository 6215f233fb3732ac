for rep in 1 2; do for g in qwen448 baseline; do for v in prod blk128 blk256 y4 nw2; do
if [ $v = prod ]; then unset RETAKE_HIP_LIB; else export RETAKE_HIP_LIB=$PWD/video-retake_amd/retake/_lib/variants/libretake_hip_$v.so; fi
for mode in "--pre-rope" ""; do
python bench.py --geometry $g $mode --no-cpu-baseline --no-extras --no-self-check --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_untimed_single_stream']
kk='prologue' if 'prologue' in k else 'unrotate_pack'
print('$g rep$rep $v %-10s: %.1f frames/s  %s %.2f us' % ('$mode', d['value'], kk, k[kk]['avg_us']))"
done; done; done; done
