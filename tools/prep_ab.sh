#!/bin/bash
# A/B of the per-layer kernels on one box: tools/prep_ab.sh <variant.so> ...   ("" = the in-tree build)
for lib in "$@"; do
  for rep in 1 2; do
    RETAKE_HIP_LIB=$lib timeout 300 python bench.py --frames 256 --steps 1 --warmup 0 --no-cpu-baseline --also-streams 0 --no-extras --report /dev/stderr 2>&1 >/dev/null < /dev/null | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_untimed_single_stream']; print('$lib'[-24:], ' '.join('%s=%.1f' % (n, k[n]['avg_us']) for n in ('unrotate_pack','rope_table','position_shift','pivotkv_select','evict_batched','commit_batched','score_finalize') if n in k))"
  done
done
