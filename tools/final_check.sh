out=gpurun_out/r08; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; grep -E "passed|failed" $out/pytest.log
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python bench.py --report $out/bench_bf16.json > $out/bench_bf16.line 2> $out/bench_bf16.err; python tools/show_bench.py $out/bench_bf16.json | grep -E "value=|real_geo|reference_r|fp32_par|fast_r|fp16|cpu_base|timed"
B="--steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-self-check --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 bench.py $B --score-rounding fast > $out/kt.log 2>&1
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); cp $f $out/baseline_fast_kernel_stats.csv; rm -rf $out/kt
