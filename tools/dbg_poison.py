"""Debugging aid: does any path of the cache read memory nobody wrote?  Every torch.empty / torch.empty_like the cache module
makes is filled with NaN (floats) / a sentinel (ints) first; a block cache with deferred re-rotation (what a sharded rank
runs) and a sequential cache compress the same chunks; V and ids must agree and contain no NaN."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/video-retake_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch
import bench as B
import retake.longvideo_cache as lc
import retake.visual_compression as vc

real_empty = torch.empty
def poisoned_empty(*a, **k):
    t = real_empty(*a, **k)
    if t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype in (torch.int64, torch.int32):
            t.fill_(-77)
        elif t.dtype == torch.uint8:
            t.fill_(0xA5)
    return t
class TorchProxy:
    def __getattr__(self, n):
        return poisoned_empty if n == "empty" else getattr(torch, n)
lc.torch = TorchProxy()

dev = torch.device("cuda:0")
L = B.FRAMES_PER_CHUNK * B.N_PATCH
layers, n_chunks = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 9
for td in (torch.bfloat16, torch.float32):
    rotary = B.Rotary(dev)
    pool = [B.pool_set(i, dev, td, projection_layout=True) for i in range(n_chunks * layers)]
    frames_all = torch.cat([B.chunk_frames(c, dev, td) for c in range(n_chunks)])[None]
    _, mask = vc.memory_bank_compress_keyframe(frames_all, n_chunks * 32, 3, sync=False)
    keep = 1568
    def build(defer, reserve):
        cache = lc.build_kvcache(B.make_cache_config(layers), reserve_tokens=reserve)
        cache.prologue_operands = "pre_rope"
        cache.defer_rerotation = defer
        for c in range(n_chunks):
            cache.keypatches_mask_chunk = mask[c * L:(c + 1) * L]
            cache.kvcache_compression = True
            pos = B.chunk_position_ids(c, dev)
            for l in range(layers):
                q0, k0, v = pool[(c * layers + l) % len(pool)]
                assert cache.update_pre_rope(q0, k0, v, l, pos, rotary, B.MROPE, query_out=real_empty(q0.shape, dtype=q0.dtype, device=dev)) is not None
            cache.after_forward()
        return cache
    seq = build(False, None)
    for reserve in (n_chunks * keep + L, None):
        blk = build(True, reserve)
        for l in range(layers):
            vs, vb = seq.value_cache[l], blk.value_cache[l]
            ps, pb = seq.position_cache[l], blk.position_cache[l]
            nan_s, nan_b = int(torch.isnan(vs.float()).sum()), int(torch.isnan(vb.float()).sum())
            bad = (vs != vb).any(-1).any(1)[0] if vs.shape == vb.shape else None
            print(f"{td} reserve={reserve} layer {l}: NaN in seq V {nan_s}, in block V {nan_b}; V equal {bool(torch.equal(vs, vb))}; ids equal {bool(torch.equal(ps, pb))}; "
                  f"K NaN seq {int(torch.isnan(seq.key_cache[l].float()).sum())} block {int(torch.isnan(blk.key_cache[l].float()).sum())}", flush=True)
print("done")
