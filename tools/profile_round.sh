#!/bin/bash
# Full round measurement on the GPU box: parity tests, bench line, rocprofv3 kernel stats, PMC passes - at BOTH geometries
# (BASELINE's synthetic one and the real Qwen2-VL one) and for the opt-in fast score arithmetic.
#   tools/profile_round.sh <out-subdir>      (results under gpurun_out/<out-subdir>/)
out=gpurun_out/${1:-round}
mkdir -p $out
export TMPDIR=/tmp
if [ -z "$PROFILE_ONLY" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1 < /dev/null
grep -E "passed|failed" $out/pytest.log
# the parity statistics the tests print (per row / per fixture), kept with the round's numbers
grep -E "^\[|^[a-z0-9_]+: |scores differ|kept tokens differ|mode vs" $out/pytest.log | grep -v "^tests/" > $out/parity_stats.txt
timeout 1500 python bench.py --report $out/bench_bf16.json > $out/bench_bf16.line 2> $out/bench_bf16.err < /dev/null
python tools/show_bench.py $out/bench_bf16.json | head -40
timeout 600 python tools/bench_ratio1.py > $out/bench_ratio1.json 2>/dev/null < /dev/null
timeout 900 python tools/bench_mallm.py > $out/bench_mallm.json 2>/dev/null < /dev/null
timeout 900 python tools/bench_llava.py > $out/bench_llava.json 2>/dev/null < /dev/null
timeout 600 python tools/bench_decode_prologue.py > $out/bench_decode.json 2>/dev/null < /dev/null
fi
# the rocprofv3 runs below use --no-self-check: the untimed self-check launches the score kernels once more per checked
# unit with gridDim.y = 1, which would mix 28x shorter launches into the per-kernel averages
B="--steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-self-check --no-extras"
kt() {  # kt <name> <bench flags...>: rocprofv3 kernel trace + stats of one timed step
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_$name -o kt --output-format csv -- python3 bench.py $B "$@" > $out/kt_$name.log 2>&1 < /dev/null
  f=$(find $out/kt_$name -name "*kernel_stats.csv" | head -1); cp $f $out/${name}_kernel_stats.csv
  t=$(find $out/kt_$name -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $t > $out/${name}_trace_summary.txt 2>&1
  rm -rf $out/kt_$name; head -12 $out/${name}_trace_summary.txt
}
kt baseline
kt qwen448 --geometry qwen448
kt qwen448_prerope --geometry qwen448 --pre-rope
kt baseline_prerope --pre-rope
kt baseline_reference --score-rounding reference
# PMC passes (counters only, own runs): HBM traffic and SQ activity on a 4-chunk video of each geometry
pmc() {  # pmc <name> <bench flags...>
  name=$1; shift
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    set -- $pass "--" "$@"; p=$1; shift; ctrs=""
    while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
    timeout 600 rocprofv3 --pmc $ctrs -d $out/${name}_$p -o $p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-self-check --no-extras "$@" > $out/${name}_$p.log 2>&1 < /dev/null
  done
  python tools/pmc_summary.py $out/${name}_pmc_hbm_traffic.csv $(find $out/${name}_fetch $out/${name}_write -name "*counter_collection.csv") 2>&1 | tail -1
  python tools/pmc_summary.py $out/${name}_pmc_sq.csv $(find $out/${name}_sq1 $out/${name}_sq2 -name "*counter_collection.csv") 2>&1 | tail -1
  rm -rf $out/${name}_fetch $out/${name}_write $out/${name}_sq1 $out/${name}_sq2
}
pmc baseline --frames 128
pmc qwen448 --geometry qwen448 --frames 256
pmc qwen448_prerope --geometry qwen448 --frames 256 --pre-rope
cat $out/baseline_pmc_hbm_traffic.csv
