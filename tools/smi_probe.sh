#!/bin/bash
# Power / shader clock while bench.py runs: samples rocm-smi every ~0.3 s, prints the samples taken under load.
#   tools/smi_probe.sh            (on the GPU box; results under gpurun_out/smi/)
mkdir -p gpurun_out/smi
( python bench.py --no-cpu-baseline --steps 6 --warmup 1 --no-self-check --report gpurun_out/smi/bench.json > gpurun_out/smi/bench.line 2>/dev/null ) &
BP=$!
: > gpurun_out/smi/smi.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | tr -s ' \t' ' ' | tr '\n' ' ' >> gpurun_out/smi/smi.txt
  echo >> gpurun_out/smi/smi.txt
  sleep 0.2
done
rocm-smi --showmaxpower 2>&1 | grep -i "Max Graphics" | tr -s ' \t' ' ' > gpurun_out/smi/cap.txt
sort -t: -k5 -n -r gpurun_out/smi/smi.txt | head -0
awk '{for(i=1;i<=NF;i++) if($i=="(W):") p=$(i+1); if (p+0 > 600) print}' gpurun_out/smi/smi.txt | head -30
echo "samples: $(wc -l < gpurun_out/smi/smi.txt)"; cat gpurun_out/smi/cap.txt; python tools/show_bench.py gpurun_out/smi/bench.json | head -1
