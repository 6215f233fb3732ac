#!/bin/bash
# A/B of score-kernel variants (env-selected) on one box.  usage: pp_ab.sh outdir "ENV=.. ENV=.." ...
out=gpurun_out/${1:-ppab}; shift
mkdir -p $out
export TMPDIR=/tmp
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 120 python tools/bench_score.py --iters 30 2>&1 < /dev/null | grep -E "score_pass|score mean"
done | tee $out/ab.txt
