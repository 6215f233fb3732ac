#!/bin/bash
# One GPU-box pass: parity tests, then a kernel trace of a short bench with a gap / per-kernel summary.
#   tools/gpu_check.sh <out-subdir> [pytest -k expression] [frames]
out=gpurun_out/${1:-chk}
kexpr=${2:-}
frames=${3:-256}
mkdir -p $out
export TMPDIR=/tmp
if [ -n "$kexpr" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q -k "$kexpr" > $out/pytest.log 2>&1 < /dev/null
else
  timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 < /dev/null
fi
tail -4 $out/pytest.log
timeout 300 rocprofv3 --kernel-trace -d $out/kt -o kt --output-format csv -- python3 bench.py --frames $frames --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --also-streams 0 > $out/kt.log 2>&1 < /dev/null
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f > $out/trace_summary.txt 2>&1 < /dev/null
head -30 $out/trace_summary.txt
rm -f $f   # the raw trace is large; the summary is what we keep
