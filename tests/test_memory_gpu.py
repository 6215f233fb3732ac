"""Device memory of the PivotKV cache: accounting, a bound on the peak, and the flush_every_layers knob.

The reference's headline claim is memory ("8x longer ... same memory", README.md:3): its cache holds the compressed rows
only (longvideo_cache.py:313-318) and its transients - [Hq, L, L] softmax tensors, two torch.cat copies of the layer -
come and go inside every update.  The build pre-allocates its transients once per chunk geometry; these tests pin how
much that is.
"""
import types

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu

Hq, Hkv, D = 28, 4, 128
SEC = [16, 24, 24]
A = synth.YARN_FACTOR4_ATTENTION_SCALING


def dev():
    return torch.device("cuda:0")


def cfg(layers, **extra):
    kw = {"compression_ratio": 0.25, "compression_method": "pivotkv", "pos_embed_reforge": True}
    kw.update(extra)
    return types.SimpleNamespace(hidden_size=Hq * D, num_hidden_layers=layers, num_attention_heads=Hq, num_key_value_heads=Hkv,
                                 longvideo_kwargs={"kvcache_compression": True, "kvcache_compression_kwargs": kw})


def ids(c, L):
    return torch.from_numpy(synth.mrope_position_ids(10 + 7 * c, L // 64, 8, 8, hw0=2)).to(dev())


def inputs(n, L, dtype, seed=0):
    g = torch.Generator(device=dev()).manual_seed(seed)
    return [tuple((1.7 * torch.randn((1, L, h, D), generator=g, device=dev())).to(dtype).transpose(1, 2) for h in (Hq, Hkv, Hkv))
            for _ in range(n)]


def run_video(cache, pool, layers, n_chunks, L, rot, route):
    call = 0
    for c in range(n_chunks):
        cache.keypatches_mask_chunk = torch.from_numpy(np.random.default_rng(c).uniform(size=L) < 0.3).to(dev())
        cache.kvcache_compression = True
        pos = ids(c, L)
        for l in range(layers):
            q, k, v = pool[call % len(pool)]
            call += 1
            if route == "prologue":
                assert cache.update_pre_rope(q.clone(), k, v, l, pos, rot, SEC) is not None
            else:
                cache.shift_temporal_ids_(pos, l)
                qr = synth.rope_forward(q.float(), pos, rot, SEC).to(q.dtype)
                kr = synth.rope_forward(k.float(), pos, rot, SEC).to(q.dtype)
                cache.update(kr, v, l, {"query_states": qr, "position_ids": pos, "rotary_emb": rot, "mrope_section": SEC})
        cache.after_forward()


@pytest.mark.parametrize("route", ["update", "prologue"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_flush_every_layers_equals_one_flush_per_chunk(dtype, route):
    """flush_every_layers = N (slot = layer mod N, a flush every N layers) builds bit for bit the cache of the default (one
    flush per chunk for all layers) - N dividing the layer count or not - with N / layers of its scratch."""
    import retake.longvideo_cache as lc

    layers, n_chunks, L = 6, 3, 640
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    pool = inputs(layers * n_chunks, L, dtype)
    ref = lc.build_kvcache(cfg(layers))
    run_video(ref, pool, layers, n_chunks, L, rot, route)
    ref_fp = ref.memory_footprint()
    scratch = lambda fp: fp["total"] - fp["cache_rows"] - fp["cache_headroom"]   # noqa: E731
    for n in (1, 2, 4, 6, 9):
        cache = lc.build_kvcache(cfg(layers, flush_every_layers=n))
        run_video(cache, pool, layers, n_chunks, L, rot, route)
        assert cache._batch.slots == min(n, layers) and cache._batch.wrap
        for l in range(layers):
            assert torch.equal(cache.key_cache[l], ref.key_cache[l]), (n, l)
            assert torch.equal(cache.value_cache[l], ref.value_cache[l]) and torch.equal(cache.position_cache[l], ref.position_cache[l])
        assert cache.num_evicted_tokens == ref.num_evicted_tokens
        fp = cache.memory_footprint()
        assert fp["cache_rows"] == ref_fp["cache_rows"]
        ratio = scratch(fp) / scratch(ref_fp)
        # per-slot scratch scales with the slots; the fp32 / per-unit score passes keep one slot-independent workspace
        assert -0.05 < ratio - min(n, layers) / layers < (0.05 if dtype != torch.float32 else 0.15), (n, ratio)
    if route == "update":   # ... and together with worker streams (scoring of each update on one of two side streams)
        cache = lc.build_kvcache(cfg(layers, flush_every_layers=2, overlap_streams=2))
        run_video(cache, pool, layers, n_chunks, L, rot, route)
        for l in range(layers):
            assert torch.equal(cache.key_cache[l], ref.key_cache[l]) and torch.equal(cache.value_cache[l], ref.value_cache[l])
            assert torch.equal(cache.position_cache[l], ref.position_cache[l])


def test_footprint_accounts_for_the_cache_and_bounds_the_peak():
    """A 6-chunk, 8-layer bf16 video at the real chunk length (L = 2304) through the product-default prologue route, with the
    capacity hint the patched forward gives (compressed video + one chunk):
      * memory_footprint() is the allocator's view: its total equals the bytes the video allocated and still holds (2 %);
      * the PEAK over the video, inputs excluded, stays within
            compressed rows + the in-flight chunk of every layer + layers x 1.25 x (Hq + 2 Hkv) L D elements + 64 MB
        (q~ + k~ + partials per slot; the reference's one-update transient at this L is Hq L^2 (4 + 2 + 2) bytes = 1.2 GB);
      * flush_every_layers = 2 cuts the scratch term by layers / 2."""
    import os

    import retake.longvideo_cache as lc

    if os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1":
        pytest.skip("torch keeps no allocator statistics without its caching allocator (memory_allocated() is 0)")
    layers, n_chunks, L, dtype = 8, 6, 2304, torch.bfloat16
    keep, es = L // 4, 2
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    pool = inputs(8, L, dtype)
    out = {}
    import gc

    for n in (0, 2):
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        base = torch.cuda.memory_allocated()
        torch.cuda.reset_peak_memory_stats()
        cache = lc.build_kvcache(cfg(layers, flush_every_layers=n), reserve_tokens=n_chunks * keep + L)
        run_video(cache, pool, layers, n_chunks, L, rot, "prologue")
        torch.cuda.synchronize()
        held = torch.cuda.memory_allocated() - base
        peak = torch.cuda.max_memory_allocated() - base
        fp = cache.memory_footprint()
        rows = layers * 2 * Hkv * D * es * n_chunks * keep
        chunk = layers * 2 * Hkv * D * es * L
        slots = n if n else layers
        scratch_bound = int(slots * 1.25 * (Hq + 2 * Hkv) * L * D * es) + (64 << 20)
        assert abs(fp["total"] - held) <= 0.02 * held, (fp["total"], held)
        assert fp["cache_rows"] == rows + layers * 3 * 8 * n_chunks * keep
        assert peak <= rows + chunk + scratch_bound, (peak, rows, chunk, scratch_bound)
        out[n] = (peak, fp["total"] - fp["cache_rows"] - fp["cache_headroom"])
        print(f"\n[memory] flush_every_layers={n}: peak {peak / 2**20:.0f} MiB = rows {rows / 2**20:.0f} + chunk {chunk / 2**20:.0f} + "
              f"scratch {out[n][1] / 2**20:.0f} MiB (+ per-call temporaries); footprint {fp['total'] / 2**20:.0f} MiB, held {held / 2**20:.0f} MiB")
        del cache
    assert out[2][1] < 0.3 * out[0][1]
    ref_transient = Hq * L * L * (4 + es + es)
    assert out[0][1] < ref_transient      # all 8 layers' scratch together stays under ONE reference update's transients
