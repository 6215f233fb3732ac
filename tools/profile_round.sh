#!/bin/bash
# Full round measurement on the GPU box: parity tests, bench line, rocprofv3 kernel stats, PMC passes.
#   tools/profile_round.sh <out-subdir>      (results under gpurun_out/<out-subdir>/)
out=gpurun_out/${1:-round}
mkdir -p $out
export TMPDIR=/tmp
if [ -z "$PROFILE_ONLY" ]; then
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 < /dev/null
grep -E "passed|failed" $out/pytest.log
timeout 900 python bench.py > $out/bench_bf16.json 2> $out/bench_bf16.err < /dev/null
python tools/show_bench.py < $out/bench_bf16.json | head -40
timeout 600 python bench.py --dtype fp32 --steps 1 --warmup 1 --no-cpu-baseline --also-streams 0 > $out/bench_fp32.json 2> $out/bench_fp32.err < /dev/null
python tools/show_bench.py < $out/bench_fp32.json | head -3
fi
# the rocprofv3 runs below use --no-self-check: the untimed self-check launches the score kernels once more per checked
# unit with gridDim.y = 1, which would mix 28x shorter launches into the per-kernel averages
# rocprofv3 kernel trace + stats of the same bench command (one timed step)
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-self-check --also-streams 0 > $out/kt.log 2>&1 < /dev/null
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
t=$(find $out/kt -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $t > $out/trace_summary.txt 2>&1; rm -f $t
head -16 $out/trace_summary.txt
# PMC passes (counters only, own runs): HBM traffic and SQ activity on a 4-chunk video
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  set -- $pass; name=$1; shift
  timeout 600 rocprofv3 --pmc $@ -d $out/$name -o $name --output-format csv -- python3 bench.py --frames 128 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-self-check --also-streams 0 > $out/$name.log 2>&1 < /dev/null
done
python tools/pmc_summary.py $out/pmc_hbm_traffic.csv $(find $out/fetch $out/write -name "*counter_collection.csv") 2>&1 | tail -1
python tools/pmc_summary.py $out/pmc_sq.csv $(find $out/sq1 $out/sq2 -name "*counter_collection.csv") 2>&1 | tail -1
rm -rf $out/fetch $out/write $out/sq1 $out/sq2 $out/kt
cat $out/pmc_hbm_traffic.csv
