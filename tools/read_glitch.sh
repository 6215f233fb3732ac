#!/bin/bash
# tools/read_glitch.sh <name> <nproc> <no_caching 0|1> [probe args]   -> gpurun_out/p2p_hunt/glitch_<name>.log
out=gpurun_out/p2p_hunt; mkdir -p $out
name=$1; np=$2; nc=$3; shift 3
port=$((20000 + RANDOM % 20000))
env PYTORCH_NO_CUDA_MEMORY_CACHING=$nc timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$np \
    --master-addr 127.0.0.1 --master-port $port tools/read_glitch_probe.py "$@" > $out/glitch_$name.log 2>&1
echo "== $name rc=$?: $(grep READ_GLITCH $out/glitch_$name.log)"
grep -h "^rank" $out/glitch_$name.log | head -4
