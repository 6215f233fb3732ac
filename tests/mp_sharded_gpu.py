#!/usr/bin/env python3
"""Multi-rank GPU check of the chunk-sharded path, launched by tests/test_hip_parity.py::test_sharded_multi_rank_rccl as
    python -m torch.distributed.run --nproc-per-node W --master-addr 127.0.0.1 --master-port P tests/mp_sharded_gpu.py
Every rank compresses its block of one small synthetic video (bench.py's deterministic tensors taken as pre-RoPE
contents, rotated at the ids each run really uses - a block's provisional ids differ from the sequential run's) through
retake.sharded.sharded_video_step - RCCL all-gathers of the distance rows, the temporal offsets and the compressed cache -
and compares the assembled cache with the cache the same rank builds sequentially on its own: ids and V exact, K within
3e-6 of the largest key (fp32).  Two shapes: chunks divisible by the world size (per-chunk overlapped gathers) and one chunk more (ragged
blocks, padded assembly at the end).  Prints MP_SHARDED_OK on rank 0.

RETAKE_TEST_TRANSPORT=p2p runs the same check over the direct peer-to-peer pushes of retake/p2p.py with a gloo control
plane and rank r on GPU r % device_count - two ranks can then share the one GPU of a test box (RCCL refuses that) - and
repeats the even case so that the landing buffers of the per-chunk pushes are reused across videos.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist


def main():
    import bench as B
    import retake.longvideo_cache as lc
    import retake.visual_compression as vc
    import synth
    from retake import sharded

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    p2p = os.environ.get("RETAKE_TEST_TRANSPORT") == "p2p"
    dev = torch.device("cuda", 0 if os.environ.get("RETAKE_TEST_ONE_GPU") == "1"
                       else int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if p2p:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        group = sharded.enable_p2p(device=dev)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    state = {}
    td, layers = torch.float32, 2
    L = B.FRAMES_PER_CHUNK * B.N_PATCH
    rotary = B.Rotary(dev)
    for n_chunks in (2 * world, 2 * world + 1) + ((2 * world, 2 * world, 2 * world) if p2p else ()):
        T = n_chunks * B.FRAMES_PER_CHUNK
        pool = [B.pool_set(i, dev, td) for i in range(n_chunks * layers)]
        def inputs(c, l, pos):   # what the model hands the cache: contents rotated at the ids in use
            q0, k0, v = pool[(c * layers + l) % len(pool)]
            return synth.rope_forward(q0, pos, rotary, B.MROPE), synth.rope_forward(k0, pos, rotary, B.MROPE), v

        # sequential single-GPU reference on this rank
        frames_all = torch.cat([B.chunk_frames(c, dev, td) for c in range(n_chunks)])[None]
        _, mask = vc.memory_bank_compress_keyframe(frames_all, T, 3, sync=False)
        seq = lc.build_kvcache(B.make_cache_config(layers))
        for c in range(n_chunks):
            seq.keypatches_mask_chunk = mask[c * L:(c + 1) * L]
            seq.kvcache_compression = True
            pos = B.chunk_position_ids(c, dev)
            for l in range(layers):
                seq.shift_temporal_ids_(pos, l)
                q, k, v = inputs(c, l, pos)
                seq.update(k, v, l, {"query_states": q, "position_ids": pos, "rotary_emb": rotary, "mrope_section": B.MROPE})
            seq.after_forward()
        # sharded
        blocks = sharded.shard_chunks(n_chunks, world)
        c0, c1 = blocks[rank]
        even = len({b - a for a, b in blocks}) == 1
        halo = c0 > 0
        parts = ([B.chunk_frames(c0 - 1, dev, td)[-1:]] if halo else []) + [B.chunk_frames(c, dev, td) for c in range(c0, c1)]
        pos_base = [B.chunk_position_ids(c, dev) for c in range(c0, c1)]
        _, (keys, values, pos) = sharded.sharded_video_step(torch.cat(parts)[None], halo, T, c0, c1, layers, pool, pos_base,
                                                            rotary, even, state=state, inputs=inputs)
        keep = max(1, int(B.RATIO * L))
        for l in range(layers):
            assert keys[l].shape[2] == n_chunks * keep, (keys[l].shape, n_chunks * keep)
            if not torch.equal(pos[l], seq.position_cache[l]):
                bad = (pos[l] != seq.position_cache[l]).reshape(-1, pos[l].shape[-1]).any(0).reshape(-1, keep).sum(1)
                raise AssertionError(f"layer {l}: ids differ; wrong ids per kept chunk {bad.tolist()}; first rows "
                                     f"{pos[l].reshape(-1, pos[l].shape[-1])[0, ::keep].tolist()} vs "
                                     f"{seq.position_cache[l].reshape(-1, pos[l].shape[-1])[0, ::keep].tolist()}")
            assert torch.equal(values[l], seq.value_cache[l]), f"layer {l}: V differs"
            # R(delta) R(p) vs R(p + delta) in fp32: the two angle roundings differ by up to an ulp of the angle, so the
            # bound scales with |k| (1.7 sigma inputs here; 1e-5 at unit scale): 3e-6 relative to the largest key
            err = (keys[l] - seq.key_cache[l]).abs().max().item()
            if err > 3e-6 * seq.key_cache[l].abs().max().item():   # say where: per kept chunk of the assembled rows
                d = (keys[l] - seq.key_cache[l]).abs().amax(dim=(0, 1, 3)).reshape(-1, keep).amax(dim=1)
                raise AssertionError(f"layer {l}: K differs by {err}; max |diff| per kept chunk {d.tolist()}")
        a, b = B.cache_checksum(keys, values, pos), B.cache_checksum([seq.key_cache[l] for l in range(layers)],
                                                                    [seq.value_cache[l] for l in range(layers)],
                                                                    seq.position_cache)
        assert a["ids_sum"] == b["ids_sum"] and a["v_bits_sum"] == b["v_bits_sum"] and a["tokens_per_layer"] == b["tokens_per_layer"]
        assert abs(a["k_abs_sum"] - b["k_abs_sum"]) <= 1e-6 * b["k_abs_sum"]
        if p2p:
            group.check()
            torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            print(f"chunks {n_chunks} on {world} rank(s): blocks {blocks}, overlapped gathers {even}: assembled == sequential", flush=True)
    if p2p:
        sharded.disable_p2p()
    dist.destroy_process_group()
    if rank == 0:
        print("MP_SHARDED_OK", flush=True)


if __name__ == "__main__":
    main()
