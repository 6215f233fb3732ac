out=gpurun_out/r12b; mkdir -p $out; export TMPDIR=/tmp
B="--steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-self-check --no-extras"
kt() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_$name -o kt --output-format csv -- python3 bench.py $B "$@" > $out/kt_$name.log 2>&1 < /dev/null
  f=$(find $out/kt_$name -name "*kernel_stats.csv" | head -1); cp $f $out/${name}_kernel_stats.csv
  t=$(find $out/kt_$name -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $t > $out/${name}_trace_summary.txt 2>&1
  rm -rf $out/kt_$name; head -14 $out/${name}_trace_summary.txt; tail -2 $out/kt_$name.log | cut -c1-300
}
kt qwen448 --geometry qwen448
kt qwen448_prerope --geometry qwen448 --pre-rope
