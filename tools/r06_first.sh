#!/bin/bash
# round 3, first GPU call: parity tests, bench line with the companions, rocprofv3 kernel stats at the real geometry
out=gpurun_out/${1:-r06a}
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1 < /dev/null
grep -E "passed|failed|rows [0-9]+:" $out/pytest.log | tail -20
timeout 900 python bench.py > $out/bench_bf16.json 2> $out/bench_bf16.err < /dev/null
tail -3 $out/bench_bf16.err
python tools/show_bench.py < $out/bench_bf16.json | head -30
timeout 600 python bench.py --geometry qwen448 --no-cpu-baseline --no-extras > $out/bench_qwen448.json 2> $out/bench_qwen448.err < /dev/null
python tools/show_bench.py < $out/bench_qwen448.json | head -30
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 bench.py --geometry qwen448 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-self-check --no-extras > $out/kt.log 2>&1 < /dev/null
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); cp $f $out/qwen448_kernel_stats.csv
rm -rf $out/kt
head -20 $out/qwen448_kernel_stats.csv
