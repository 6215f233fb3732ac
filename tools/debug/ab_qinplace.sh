for rep in 1 2; do for g in qwen448 baseline; do for o in 1 0; do
python bench.py --geometry $g --pre-rope --no-cpu-baseline --steps 2 --warmup 1 --cache-option score_queries_in_place=$o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_untimed_single_stream']; t=d['kernels_timed_region']
print('$g rep$rep in_place=$o: %.1f frames/s  p1 %.1f p2 %.1f prologue %.2f us' % (d['value'], t['score_pass1']['avg_us'], t['score_pass2']['avg_us'], k['prologue']['avg_us']))"
done; done; done
