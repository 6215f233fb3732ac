#!/bin/bash
# The three discriminating runs of VERDICT r5 #4 (profiles/r15_p2p_hunt.log): eight ranks on ONE GPU over the p2p transport,
# WITHOUT torch's caching allocator, sequence 16/17-chunk set -> 64 -> 65 (tests/mp_sharded_gpu.py).
#   tools/p2p_hunt.sh <name> [ENV=VALUE ...]     one run, log to gpurun_out/p2p_hunt/<name>.log
out=gpurun_out/p2p_hunt; mkdir -p $out
name=$1; shift
port=$((20000 + RANDOM % 20000))
t0=$(date +%s)
env PYTORCH_NO_CUDA_MEMORY_CACHING=1 RETAKE_TEST_TRANSPORT=p2p RETAKE_TEST_ONE_GPU=1 RETAKE_TEST_MORE_CASES="bf16:64,65" \
    RETAKE_VERIFY_POOL_WATCH=1 "$@" \
    timeout -k 10 ${HUNT_TIMEOUT:-420} python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port $port \
    tests/mp_sharded_gpu.py > $out/$name.log 2>&1
rc=$?
echo "== $name ($*): rc=$rc in $(( $(date +%s) - t0 )) s; $(grep -c 'assembled == sequential' $out/$name.log) videos ok; MP_SHARDED_OK: $(grep -c MP_SHARDED_OK $out/$name.log)"
grep -h "POOLWATCH\|  pool set\|  ranges mapped\|AssertionError\|timed out" $out/$name.log | cut -c1-700 | head -12
