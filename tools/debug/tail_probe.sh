for n in 21 23 24 25 27 28 30 32 33; do
python bench.py --geometry qwen448 --layers $n --no-extras --no-cpu-baseline --no-self-check --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_timed_region']
n=$n
print(n, 'layers: pass1 %.1f us (%.2f per unit, %d WGs = %.2f rounds of 768)  pass2 %.1f us (%.2f per unit)' % (k['score_pass1']['avg_us'], k['score_pass1']['avg_us']/n, 252*n, 252*n/768, k['score_pass2']['avg_us'], k['score_pass2']['avg_us']/n))"
done
