#!/usr/bin/env python3
"""bench.py — frames/s through DPSelect + PivotKV at 2048 frames on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--dtype bf16|fp32] [--frames 2048]

One "step" = one pass of the hot path over one synthetic 2048-frame video, inputs resident in HBM:
    DPSelect on [1, 2048, 196, 1280] frame embeddings (shipped default: ratio 1.0, patch_sync False)
    + PivotKVCache.update for every (chunk, layer): 64 chunks x 28 layers, L = 6272 tokens per chunk,
      Hq 28 / Hkv 4 / D 128, ratio 0.25 (4x KV compression), pos_embed_reforge, M-RoPE [16,24,24],
      YaRN attention_scaling 1.1386, key-patch mask from DPSelect.
All of it goes through the product's plugin surface (retake.visual_compression /
retake.longvideo_cache), i.e. the C ABI of libretake_hip.so.  N > 1: one process per GPU over RCCL,
the video's frame chunks are sharded across ranks (strong scaling), see retake/sharded.py.

Prints ONE short JSON line (rank 0, stdout, < 4 KB: the contract keys + one number per extra measurement); the full
report (per-kernel tables, companions, memory split) goes to bench_report.json and stderr (`contract_line` / `emit`).
`roofline` is for the dominant kernel, measured with HIP events on the launch stream inside the timed region
(rtk_profile_*); `cpu_baseline` times the CPU oracle (oracle/, test infrastructure) on a bounded sample of the same
workload on this host's cores.
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}  # dense peaks, MI355X_MICROARCH.md
TORCH_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}

Hq, Hkv, D = 28, 4, 128
N_PATCH, C_EMB = 196, 1280
GRID_H, GRID_W = 14, 14
FRAMES_PER_CHUNK = 32          # temporal rows of the frame bank per PivotKV chunk
FRAMES_PER_ROW = 1             # video frames one row of the frame bank stands for
LAYERS = 28
SCORE_ROUNDING = "fp32"

# name -> (patch positions N, channels C, token grid h x w, bank rows per chunk, video frames per bank row, label)
GEOMETRIES = {
    # BASELINE.json's synthetic shape: 2048 x (14 x 14) x 1280 embeddings, chunk L = 32 x 196 = 6272 (= LLaVA-Video's chunk)
    "baseline": (196, 1280, 14, 14, 32, 1, "BASELINE.json synthetic geometry"),
    # the real Qwen2-VL-7B pipeline at longsize_resolution 448, 16:9 (cal_flops.py:8,47, qwen2_vl.py:477-491):
    # 32 x 18 patches -> 16 x 9 merged tokens per temporal grid, C = 3584 after the merger, one grid = 2 frames,
    # chunk = 16 grids x 144 = 2304 tokens; a 2048-frame video = 1024 grids = 64 chunks
    "qwen448": (144, 3584, 9, 16, 16, 2, "real Qwen2-VL geometry (448 px, 16:9)"),
}


def set_geometry(name: str):
    """Switches the module-level shape constants (retake/sharded.py reads them through this module too)."""
    global N_PATCH, C_EMB, GRID_H, GRID_W, FRAMES_PER_CHUNK, FRAMES_PER_ROW
    N_PATCH, C_EMB, GRID_H, GRID_W, FRAMES_PER_CHUNK, FRAMES_PER_ROW, _ = GEOMETRIES[name]
    assert GRID_H * GRID_W == N_PATCH
RATIO = 0.25
MROPE = [16, 24, 24]
A_SCALE = 0.1 * math.log(4.0) + 1.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp16"])
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--geometry", default="baseline", choices=sorted(GEOMETRIES),
                    help="shape of the HEADLINE measurement (the contract line is 'baseline' = BASELINE configs[2])")
    ap.add_argument("--score-rounding", default="fp32", choices=["fp32", "reference", "fast"],
                    help="PivotKV bf16 score arithmetic of the headline (retake.longvideo_cache score_rounding)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed-for-`value` companions: real_geometry, reference_rounding, fp32_parity_dtype")
    ap.add_argument("--layers", type=int, default=LAYERS)
    ap.add_argument("--pool", type=int, default=48, help="distinct resident (q,k,v) sets cycled over the calls")
    ap.add_argument("--streams", type=int, default=0, help="worker HIP streams per cache (0 = everything on one stream)")
    ap.add_argument("--also-streams", type=int, default=0,
                    help="after the contract measurement, time the same steps again with this many worker streams "
                         "(reported under 'overlap'; 0 = skip)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "p2p", "host"],
                    help="N > 1: torch.distributed collectives over RCCL, or the direct peer-to-peer pushes of retake/p2p.py; "
                         "'host' (tests: the collective-transport code path with several ranks on ONE GPU, which RCCL refuses) "
                         "stages every exchange through the host over gloo")
    ap.add_argument("--pre-rope", action="store_true",
                    help="measure the attention patch's fused prologue instead (PivotKVCache.update_pre_rope on pre-RoPE "
                         "projections in the projection layout); not the contract line")
    ap.add_argument("--cache-option", action="append", default=[], metavar="KEY=VALUE",
                    help="extra kvcache_compression_kwargs entry of the measured cache (A/B runs), e.g. score_queries_in_place=0")
    ap.add_argument("--report", default=None, metavar="PATH",
                    help="where the full report goes (default: bench_report.json at the repo root and in gpurun_out/)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-self-check", action="store_true")
    ap.add_argument("--cpu-sample-updates", type=int, default=10,
                    help="timed repetitions of the CPU PivotKV update (~0.7-1 s each on the GPU box's host: ~10 s of CPU work)")
    return ap.parse_args()


class Rotary:
    """inv_freq * position rotary module with YaRN attention_scaling (what HF's rotary modules compute)."""

    def __init__(self, device):
        self.inv_freq = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))).to(device)
        self.attention_scaling = A_SCALE

    def __call__(self, x, position_ids):
        inv = self.inv_freq[None, None, :, None].expand(3, position_ids.shape[1], -1, 1)
        freqs = (inv @ position_ids[:, :, None, :].float()).transpose(2, 3)
        emb = torch.cat((freqs, freqs), dim=-1)
        return (emb.cos() * self.attention_scaling).to(x.dtype), (emb.sin() * self.attention_scaling).to(x.dtype)


OVERLAP_STREAMS = 0
CACHE_EXTRA = {}               # build-specific cache options of a companion measurement (the headline sets none)
MAX_INPUT_LENGTH = 32000       # configs/retake_demo.yaml:23


def cache_kwargs():
    """kvcache_compression_kwargs of the measured cache: the reference's own YAML keys (configs/retake_demo.yaml:18-24)
    plus the `compression_ratio` its dynamic rule writes into the same dict before the cache is built
    (qwen2_vl.py:553-557; here the 4x of BASELINE configs[2]).  Build-specific options appear only when a flag / a
    companion asks for them - the headline runs on the product defaults."""
    kw = {"dynamic_compression_ratio": True, "compression_method": "pivotkv", "pos_embed_reforge": True,
          "max_input_length": MAX_INPUT_LENGTH, "compression_ratio": RATIO}
    if SCORE_ROUNDING != "fp32":
        kw["score_rounding"] = SCORE_ROUNDING
    if OVERLAP_STREAMS:
        kw["overlap_streams"] = OVERLAP_STREAMS
    kw.update(CACHE_EXTRA)
    return kw


def make_cache_config(layers):
    return types.SimpleNamespace(
        hidden_size=Hq * D, num_hidden_layers=layers, num_attention_heads=Hq, num_key_value_heads=Hkv,
        longvideo_kwargs={"kvcache_compression": True, "kvcache_compression_kwargs": cache_kwargs()})


def chunk_position_ids(c, device):
    L = FRAMES_PER_CHUNK * N_PATCH
    t = torch.arange(FRAMES_PER_CHUNK, device=device).repeat_interleave(N_PATCH) + FRAMES_PER_CHUNK * c + 16
    h = torch.arange(GRID_H, device=device).repeat_interleave(GRID_W).repeat(FRAMES_PER_CHUNK) + 16
    w = torch.arange(GRID_W, device=device).repeat(GRID_H * FRAMES_PER_CHUNK) + 16
    return torch.stack([t, h, w]).view(3, 1, L)


def _settled(dev, *temporaries):
    """Without torch's caching allocator (PYTORCH_NO_CUDA_MEMORY_CACHING=1, a debugging mode) a dropped temporary is
    hipFree'd at once, while the kernels that read it may still be queued - with eight rank processes on one GPU they
    queue for long, and a generated input then occasionally holds other numbers than its seed says (seen: one k tensor
    of 390, on one rank).  In that mode the generators wait for the device before they let go of a temporary."""
    if os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1":
        torch.cuda.synchronize(dev)
    del temporaries


def chunk_frames(c, dev, tdtype):
    """The 32 frame embeddings of frame chunk c: a function of c alone, so that any rank of a sharded run regenerates
    exactly the frames (and the halo frame) the single-GPU run sees."""
    g = torch.Generator(device=dev).manual_seed(5000 + c)
    x = torch.randn((FRAMES_PER_CHUNK, N_PATCH, C_EMB), generator=g, device=dev, dtype=torch.float32)
    out = x.to(tdtype)
    _settled(dev, x)
    return out


def pool_set(i, dev, tdtype, projection_layout=False):
    """Resident (q, k, v) set i; update (chunk c, layer l) of ANY run uses set (c * layers + l) % pool size.
    projection_layout: the memory layout q_proj / k_proj / v_proj produce ([1, L, H*D], handed on as the transposed
    [1, H, L, D] view, qwen2_vl.py:55-57) instead of head-major tensors."""
    g = torch.Generator(device=dev).manual_seed(9000 + i)
    L = FRAMES_PER_CHUNK * N_PATCH
    out = []
    for h in (Hq, Hkv, Hkv):
        x = torch.randn((1, L, h, D) if projection_layout else (1, h, L, D), generator=g, device=dev, dtype=torch.float32)
        y = 1.7 * x
        z = y.to(tdtype)
        _settled(dev, x, y)
        out.append(z.transpose(1, 2) if projection_layout else z)
    return tuple(out)


def cache_checksum(keys, values, pos):
    """Order-sensitive-free fingerprints of an assembled compressed cache, comparable between the single-GPU run and the
    sharded runs (same synthetic inputs by construction): token count, sum of the position ids (exact), sum of the V bit
    patterns (kept V rows are exact copies), sum of |K| (the sharded path re-rotates block by block: equal to rounding)."""
    ids = sum(int(p.sum().item()) for p in pos)
    vb = 0
    for v in values:
        iv = v.contiguous().view(torch.int16 if v.element_size() == 2 else torch.int32)
        vb += int(iv.to(torch.int64).sum().item())
    kabs = sum(float(k.float().abs().sum(dtype=torch.float64).item()) for k in keys)
    return {"tokens_per_layer": int(keys[0].shape[2]), "layers": len(keys), "ids_sum": ids, "v_bits_sum": vb,
            "k_abs_sum": kabs}


NO_VISUAL_COMPRESSION = False   # companion `no_keypatch_mask`: the reference's videomme config (visual_compression: False)


def run_video(frames, pool, masks, pos_base, rotary, layers, tdtype, pre_rope=False):
    """One step on one GPU: DPSelect + all (chunk, layer) PivotKV updates.  Returns retained tokens.
    pre_rope: the pool holds PRE-RoPE projections and every update is the attention patch's fused prologue
    (PivotKVCache.update_pre_rope: continuity shift + RoPE of q / k + append + scoring operands in one kernel).  In the
    model the rotated queries overwrite q_proj's fresh output; the pool's tensors are reused, so here they go to a
    scratch tensor of the same layout instead (rotating in place would scale a set by attention_scaling per use)."""
    import retake.longvideo_cache as lc
    import retake.visual_compression as vc

    T = frames.shape[1]
    if NO_VISUAL_COMPRESSION:   # configs/qwen2_vl/retake_qwen2-vl_videomme.yaml:6-17: no DPSelect, hence no key-patch mask
        out, mask = frames, None
    else:
        out, mask = vc.memory_bank_compress_keyframe(frames, T, 3, sync=False)
    n_chunks = T // FRAMES_PER_CHUNK
    L = FRAMES_PER_CHUNK * N_PATCH
    # capacity hint, as the patched model forward gives it (_prefill.expected_cache_tokens): compressed video + one chunk
    cache = lc.build_kvcache(make_cache_config(layers), reserve_tokens=n_chunks * max(1, int(RATIO * L)) + L)
    retained = 0
    call = 0
    q_rot = torch.empty_like(pool[0][0]) if pre_rope else None
    for c in range(n_chunks):
        cache.keypatches_mask_chunk = mask[c * L:(c + 1) * L] if mask is not None else None
        cache.kvcache_compression = True
        pos = pos_base[c].clone()
        for layer in range(layers):
            q, k, v = pool[call % len(pool)]
            call += 1
            if pre_rope:
                if cache.update_pre_rope(q, k, v, layer, pos, rotary, MROPE, query_out=q_rot) is None:
                    raise RuntimeError("update_pre_rope declined a video chunk of the benchmark geometry")
                continue
            # what the attention patch does (qwen2_vl.py:68-73): the ids tensor is shared by the layers of
            # the chunk and shifted in place, on the device (no host sync)
            cache.shift_temporal_ids_(pos, layer)
            # ("shift_next_position_ids": this build's Qwen2-VL patch sets it - retake/qwen2_vl.py - because it shifts the
            # shared ids tensor in place for every layer anyway; the next layer's shift may then ride in this launch)
            kw = {"query_states": q, "position_ids": pos, "rotary_emb": rotary, "mrope_section": MROPE,
                  "shift_next_position_ids": True}
            cache.update(k, v, layer, kw)
        cache.after_forward()
        retained += layers * max(1, int(RATIO * L))
    return retained, cache, mask


def self_check(cache, pool, mask, n_chunks, layers, rotary):
    """Untimed, after the timed region: the last chunk's first / middle / last layer as left by the chunk-batched
    launches (gridDim.y = layers) against one-unit launches of rtk_pivotkv_score + rtk_pivotkv_select on the same
    inputs — score, kept indices and new ids bitwise — and the kept V rows against a torch gather.  Raises on mismatch,
    so a run whose batched kernels mis-stride never prints a number."""
    import unit_check as uc

    L = FRAMES_PER_CHUNK * N_PATCH
    keep = max(1, int(RATIO * L))
    c = n_chunks - 1
    lay = sorted({0, layers // 2, layers - 1})
    inputs = {l: pool[(c * layers + l) % len(pool)][:2] for l in lay}
    masks = {l: mask[c * L:(c + 1) * L] for l in lay}
    res = uc.check_batch_against_units(cache, lay, inputs, masks, keep, rotary.inv_freq, A_SCALE, MROPE)
    low = []
    for l, _, idx in res:
        low.append(float((idx < keep).float().mean()))
        v = pool[(c * layers + l) % len(pool)][2]
        if not torch.equal(cache.value_cache[l][0, :, -keep:], v[0][:, idx]):
            raise AssertionError(f"self-check: layer {l} kept V rows are not copies of the selected rows")
        if cache.key_cache[l].shape[2] != n_chunks * keep:
            raise AssertionError(f"self-check: layer {l} cache length {cache.key_cache[l].shape[2]} != {n_chunks * keep}")
    return {"status": "ok", "chunk": c, "layers": lay, "staged_fraction": sum(low) / len(low),
            "checked": "batched score / keep_idx / new ids == one-unit launches (bitwise); kept V rows == gather"}


def reference_peak_formula(L, P_tokens, es):
    """Device bytes the REFERENCE holds at the peak of one PivotKVCache.update at chunk length L with P_tokens cached rows in
    the layer (formula, not a measurement: the reference cannot run here).  longvideo_cache.py:238 torch.cat of the layer
    (K and V, [P + L] rows, the old [P] rows still alive), :260-262 repeat_kv copy of the un-rotated K and the [Hq, L, L]
    logits in the model dtype, :265-267 their fp32 softmax and its cast back, while q~ / k~ (:250-259) are alive."""
    kv_row = 2 * Hkv * D * es
    terms = {
        "layer_cache_old_rows": kv_row * P_tokens,
        "torch_cat_new_layer_cache": kv_row * (P_tokens + L),
        "unrotated_q_k": (Hq + Hkv) * L * D * es,
        "repeat_kv_copy": Hq * L * D * es,
        "logits_model_dtype": Hq * L * L * es,
        "softmax_fp32": Hq * L * L * 4,
        "softmax_cast_model_dtype": Hq * L * L * es,
    }
    terms["total"] = sum(terms.values())
    return terms


def memory_block(cache, peak_alloc, peak_reserved, resident_inputs, dpselect_out_bytes, layers, L, n_chunks, es):
    """The `memory` object of the contract line: the allocator's peak over the timed region and what the product's share
    of it is made of (PivotKVCache.memory_footprint() at the end of the last video: every scratch buffer is allocated on
    the first chunk and lives as long as the cache), beside the reference's peak by formula."""
    fp = cache.memory_footprint()
    keep = max(1, int(RATIO * L))
    in_flight = 2 * layers * Hkv * L * D * es
    ref = reference_peak_formula(L, (n_chunks - 1) * keep, es)
    # + the other layers' compressed rows + DPSelect's output, which the reference materialises as well (:138 / :173)
    ref_total = ref["total"] + (layers - 1) * 2 * Hkv * D * es * n_chunks * keep + dpselect_out_bytes
    product_peak = peak_alloc - resident_inputs
    return {
        "peak_allocated_bytes": peak_alloc, "peak_reserved_bytes": peak_reserved,
        "resident_inputs_bytes": resident_inputs,
        "product_peak_bytes": product_peak,
        "note": "peak = torch.cuda.max_memory_allocated() over the timed region; resident inputs = the bench's own frame bank, "
                "(q, k, v) pool and id tensors (a model would hold one chunk's activations instead); product peak = the "
                "difference: DPSelect's outputs + the cache with its scratch",
        "split_bytes": {
            "compressed_cache_rows": fp["cache_rows"],
            "in_flight_chunk_all_layers": in_flight,
            "cache_headroom_beyond_the_chunk": fp["cache_headroom"] - in_flight,
            "k_unrotated": fp["k_unrotated"], "score_operands_q": fp["score_operands"],
            "score_partials": fp["score_partials"], "selection_scratch": fp["selection"], "staging": fp["staging"],
            "deferred_queries": fp["deferred_queries"], "worker_scratch": fp["worker_scratch"],
            "dpselect_outputs": dpselect_out_bytes,
        },
        "cache_footprint_total_bytes": fp["total"],
        "scratch_over_cache_rows": (fp["total"] - fp["cache_rows"] - in_flight) / max(1, fp["cache_rows"]),
        "reference_peak_by_formula": {"bytes": ref_total, "one_update_terms": ref,
                                      "note": "last chunk of the video, one layer inside update + the other layers' compressed "
                                              "rows + DPSelect's output copy; the [Hq, L, L] tensors dominate"},
        "product_peak_over_reference_formula": product_peak / ref_total,
    }


def profile_key():
    """Which committed PMC profile describes the kernels of the current configuration (tools/profile_round.sh names)."""
    geo = [g for g, v in GEOMETRIES.items() if v[:2] == (N_PATCH, C_EMB)][0]
    return geo + ("_fast" if SCORE_ROUNDING == "fast" else "")


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the newest committed PMC profile of the current geometry / score arithmetic
    (profiles/rNN_<geometry>[_fast]_pmc_hbm_traffic.csv: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    FETCH_SIZE x2 per MI355X_MICROARCH.md).  bench.py cannot collect counters itself; the profile is of the same
    kernels on the same shapes, whole chunks (28 units) per launch."""
    import csv
    import glob

    if SCORE_ROUNDING == "reference":
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{profile_key()}_pmc_hbm_traffic.csv")))
    if not files:
        return None, None
    for row in csv.DictReader(open(files[-1])):
        if kernel_substr in row["kernel"] and row.get("hbm_total_MB"):
            return float(row["hbm_total_MB"]) * 1e6, os.path.relpath(files[-1], ROOT)
    return None, None


def score_roofline(kern, dtype, L, T, n_updates):
    """roofline object of the dominant kernel (one of the two score passes): algorithmic flops per launch (one
    Q K^T per (layer, chunk) unit, SURVEY §8(d), times the units one launch processes — the cache batches all
    layers of a chunk into one launch) / average launch duration from the HIP events, against the dense MFMA peak."""
    dom = max(("score_pass1", "score_pass2"), key=lambda k: kern[k]["total_ms"])
    units_per_launch = n_updates / kern[dom]["launches"]
    flops = 2.0 * Hq * L * L * D * units_per_launch
    avg_s = kern[dom]["avg_us"] * 1e-6
    peak = MFMA_PEAK_TFLOPS[dtype]
    # the profile's launches cover 28 units; "score_pass1_dma_kernel<2, true, 2>" is the fast mode's fix-up launch
    traffic, src = pmc_traffic(dom + "_dma_kernel") if dtype == "bf16" and units_per_launch == LAYERS else (None, None)
    return {"kernel": dom, "bound": "mfma", "achieved": flops / avg_s / 1e12, "peak": peak, "unit": "TFLOP/s",
            "frac": flops / avg_s / 1e12 / peak, "traffic": traffic, "traffic_source": src,
            "units_per_launch": units_per_launch, "algorithmic_flops_per_launch": flops, "avg_launch_us": kern[dom]["avg_us"]}


def hbm_achievable(dev):
    """What a device-to-device copy reaches on this GPU, beside the nominal 8 TB/s every HBM roofline fraction above
    is quoted against (state both): a 2 GiB buffer, far beyond L2 + MALL, timed with HIP events outside
    the timed region; read + write bytes counted.  Two copies: the library's non-temporal 16-byte-vector kernel
    (rtk_profile_copy - the float4 copy MI355X_MICROARCH.md quotes 6.29 TB/s for) and torch's copy_."""
    import retake._native as nv

    n = 1 << 30   # bf16 elements: 2 GiB
    src = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    dst = torch.empty_like(src)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 2 * n * 2 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9

    nt = timed(lambda: nv.check(nv.lib.rtk_profile_copy(nv.ptr(dst), nv.ptr(src), 2 * n, nv.stream()), "rtk_profile_copy"))
    tc = timed(lambda: dst.copy_(src))
    # the product's own streaming copy on the same buffer: DPSelect's frame gather at ratio 1 (identity indices), rows of
    # 2560 bytes like the headline's frame bank
    rowb, N_ = 2560, 196
    Tg = (2 * n) // (rowb * N_)
    idx = torch.arange(Tg, dtype=torch.int64, device=dev)
    ga = timed(lambda: nv.check(nv.lib.rtk_gather_frames(nv.ptr(src), Tg, N_, rowb // 2, nv.RTK_BF16, nv.ptr(idx), Tg, 1, nv.ptr(dst),
                                                         nv.stream()), "rtk_gather_frames")) * (Tg * N_ * rowb) / (2 * n)
    del src, dst
    best = max(nt, tc, ga)
    return {"copy_GBps": best, "frac_of_nominal": best / HBM_PEAK_GBS, "nt_vector_copy_GBps": nt, "torch_copy_GBps": tc,
            "gather_frames_identity_GBps": ga, "guide_float4_copy_GBps": 6290.0,
            "note": "2 GiB device copy, read + write bytes; best of the library's non-temporal 16-byte copy (one contiguous 16 KiB "
                    "piece per workgroup; tools/ubench/copy_variants.hip), torch copy_ and the product's frame gather with "
                    "identity indices"}


def cpu_baseline(args, frames_cpu_sample, n_updates):
    """Times the CPU oracle on a bounded sample and extrapolates to the full workload."""
    from oracle import oracle as orc
    import synth

    cores = orc.num_threads()
    L = FRAMES_PER_CHUNK * N_PATCH
    Ts = frames_cpu_sample.shape[1]

    def best_of(fn, reps=3):
        """BASELINE.md §3 protocol: one untimed warm-up, then the best of `reps` timed runs."""
        fn()
        best = float("inf")
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    t_dp = best_of(lambda: orc.dpselect(frames_cpu_sample, Ts, 3, False))
    inv_f = synth.inv_freq(D)
    rot = synth.RotaryStub(inv_f, A_SCALE)
    q0, k0, v = synth.qkv_chunk(123, Hq, Hkv, L, D)
    pos = synth.mrope_position_ids(16, FRAMES_PER_CHUNK, GRID_H, GRID_W, hw0=16)
    q = synth.rope_forward(torch.from_numpy(q0), torch.from_numpy(pos), rot, MROPE).numpy()
    k = synth.rope_forward(torch.from_numpy(k0), torch.from_numpy(pos), rot, MROPE).numpy()

    def one_update():
        oc = orc.OraclePivotKV(Hq, Hkv, D, RATIO, True)
        oc.update(k, v, 0, q=q, position_ids=pos, rotary=rot, mrope_section=MROPE)

    reps_up = max(1, n_updates)
    t_up = best_of(one_update, reps_up)
    rows = args.frames // FRAMES_PER_ROW
    n_chunks = rows // FRAMES_PER_CHUNK
    total = t_dp * (rows / Ts) + t_up * n_chunks * args.layers
    return {"value": args.frames / total, "unit": "frames/s", "cores": cores, "kind": "port",
            # (short: the driver's record truncates strings; the long form is `sample_detail` in the report)
            "sample": f"oracle C+OpenMP fp32: DPSelect {Ts}/{rows} rows + 1 PivotKV update L={L} (best of {reps_up}) "
                      f"x {n_chunks}x{args.layers} updates",
            "sample_detail": f"oracle/ (C+OpenMP, fp32): DPSelect on {Ts} of {rows} bank rows ({t_dp:.2f} s) + "
                             f"one PivotKV update at L={L} ({t_up:.2f} s, one warm-up then best of {reps_up}; DPSelect: best of 3); "
                             f"extrapolated to {n_chunks}x{args.layers} updates",
            "dpselect_s_per_2048": t_dp * (rows / Ts), "pivotkv_update_s": t_up}


def companion_measurement(dev, frames_total, layers, dtype, steps, warmup, pool_n, geometry=None, score_rounding="fp32",
                          warmup_chunks=None, pre_rope=False, cache_extra=None, time_all_kernels=False, no_visual=False):
    """The same step measured again in another configuration, AFTER (and outside) the timed region `value` comes from:
    another geometry, another score arithmetic or the parity dtype.  Same protocol in small: resident inputs, `warmup`
    untimed steps (optionally shortened to `warmup_chunks` chunks - enough to build every buffer and touch every
    kernel), then `steps` timed steps between synchronisations, HIP events around the two score passes.  Returns
    {value, ms_per_step, roofline, ...}; restores the module's geometry / rounding afterwards."""
    global SCORE_ROUNDING, N_PATCH, C_EMB, GRID_H, GRID_W, FRAMES_PER_CHUNK, FRAMES_PER_ROW, CACHE_EXTRA, NO_VISUAL_COMPRESSION
    import retake._native as nv

    saved_geo = (N_PATCH, C_EMB, GRID_H, GRID_W, FRAMES_PER_CHUNK, FRAMES_PER_ROW)
    saved_round, saved_extra, saved_nv = SCORE_ROUNDING, CACHE_EXTRA, NO_VISUAL_COMPRESSION
    try:
        if geometry is not None:
            set_geometry(geometry)
        SCORE_ROUNDING = score_rounding
        CACHE_EXTRA = dict(cache_extra or {})
        NO_VISUAL_COMPRESSION = bool(no_visual)
        tdtype = TORCH_DTYPE[dtype]
        rows = frames_total // FRAMES_PER_ROW
        n_chunks = rows // FRAMES_PER_CHUNK
        L = FRAMES_PER_CHUNK * N_PATCH
        frames = torch.cat([chunk_frames(c, dev, tdtype) for c in range(n_chunks)])[None]
        pool = [pool_set(i, dev, tdtype, projection_layout=pre_rope) for i in range(min(pool_n, n_chunks * layers))]
        pos_base = [chunk_position_ids(c, dev) for c in range(n_chunks)]
        rotary = Rotary(dev)
        for _ in range(warmup):
            fr = frames if warmup_chunks is None else frames[:, : warmup_chunks * FRAMES_PER_CHUNK]
            run_video(fr, pool, None, pos_base, rotary, layers, tdtype, pre_rope)
        ids = nv.profile_kernel_ids()
        nv.check(nv.lib.rtk_profile_reset(), "profile_reset")
        nv.check(nv.lib.rtk_profile_enable_mask((1 << ids["score_pass1"]) | (1 << ids["score_pass2"])), "profile_enable")
        gc.collect()
        torch.cuda.synchronize()
        resident_inputs = torch.cuda.memory_allocated()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        cache = kp_mask = None
        step_ms = []
        for _ in range(steps):
            # the previous video's cache goes first (as after a finished `generate`): every step then reuses the blocks
            # the warm-up left in torch's allocator; with both alive, step 2 took fresh hipMallocs - 40-120 ms on a
            # device whose memory the previous companion had just released
            cache = kp_mask = None
            ts = time.perf_counter()
            _, cache, kp_mask = run_video(frames, pool, None, pos_base, rotary, layers, tdtype, pre_rope)
            torch.cuda.synchronize()
            step_ms.append((time.perf_counter() - ts) * 1e3)
        dt = time.perf_counter() - t0
        cache.check()
        es_ = 4 if dtype == "fp32" else 2
        mem = memory_block(cache, torch.cuda.max_memory_allocated(), torch.cuda.max_memory_reserved(), resident_inputs,
                           0 if NO_VISUAL_COMPRESSION else frames.numel() * frames.element_size() + 5 * rows * N_PATCH, layers, L,
                           n_chunks, es_)
        nv.check(nv.lib.rtk_profile_enable(0), "profile_enable")
        kern = {k: {"launches": n, "avg_us": ms / n * 1e3, "total_ms": ms} for k, (n, ms) in nv.profile_read().items()}
        kern_all = None
        if time_all_kernels:   # untimed extra pass with every kernel bracketed: the per-update share of the step
            cache = kp_mask = None
            nv.check(nv.lib.rtk_profile_reset(), "profile_reset")
            nv.check(nv.lib.rtk_profile_enable(1), "profile_enable")
            _, cache, kp_mask = run_video(frames, pool, None, pos_base, rotary, layers, tdtype, pre_rope)
            torch.cuda.synchronize()
            nv.check(nv.lib.rtk_profile_enable(0), "profile_enable")
            kern_all = {k: {"launches": n, "avg_us": ms / n * 1e3, "total_ms": ms} for k, (n, ms) in nv.profile_read().items()}
        check = None
        if dtype in ("bf16", "fp16") and score_rounding == "fp32" and not pre_rope and kp_mask is not None:
            check = self_check(cache, pool, kp_mask, n_chunks, layers, rotary)["status"]
        keep = max(1, int(RATIO * L))
        assert cache.key_cache[0].shape[2] == n_chunks * keep
        del cache
        res = {"value": frames_total * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "step_ms": step_ms,
               "steps": steps,
               "warmup": warmup, "dtype": dtype, "score_rounding": score_rounding,
               "retained_kv_tokens_per_s": n_chunks * layers * keep * steps / dt,
               "config": {"frames": frames_total, "bank": [1, rows, N_PATCH, C_EMB], "chunks": n_chunks, "layers": layers,
                          "chunk_tokens": L, "keep": keep,
                          "key_patch_mask_rate": float(kp_mask.float().mean().item()) if kp_mask is not None else 0.0},
               "kernels_timed_region": kern,
               "roofline": score_roofline(kern, dtype, L, rows, n_chunks * layers * steps)}
        res["config"]["cache_kwargs"] = cache_kwargs()
        res["memory"] = {k: mem[k] for k in ("product_peak_bytes", "split_bytes", "scratch_over_cache_rows",
                                             "product_peak_over_reference_formula")}
        res["memory"]["reference_peak_by_formula_bytes"] = mem["reference_peak_by_formula"]["bytes"]
        if pre_rope:
            res["config"]["update_call"] = "PivotKVCache.update_pre_rope (pre-RoPE projections, [L, H*D] layout)"
        if kern_all is not None:
            res["kernels_untimed_single_stream"] = kern_all
            per_update = sum(v["total_ms"] for k, v in kern_all.items()
                             if k in ("prologue", "unrotate_pack", "position_shift", "append", "rope_table"))
            total = sum(v["total_ms"] for k, v in kern_all.items() if not k.startswith(("dpselect", "gather_frames")))
            res["per_update_kernels_share_of_gpu_time"] = per_update / total if total else None
        if check is not None:
            res["self_check"] = check
        return res
    finally:
        N_PATCH, C_EMB, GRID_H, GRID_W, FRAMES_PER_CHUNK, FRAMES_PER_ROW = saved_geo
        SCORE_ROUNDING, CACHE_EXTRA, NO_VISUAL_COMPRESSION = saved_round, saved_extra, saved_nv
        torch.cuda.empty_cache()


class PlainRotary:
    """inv_freq * position with YaRN attention_scaling for [1, L] ids (what Qwen2's rotary module computes; LLaVA-Video)."""

    def __init__(self, device):
        self.inv_freq = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))).to(device)
        self.attention_scaling = A_SCALE

    def __call__(self, x, position_ids):
        freqs = position_ids[:, :, None].float() * self.inv_freq[None, None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        return (emb.cos() * self.attention_scaling).to(x.dtype), (emb.sin() * self.attention_scaling).to(x.dtype)


def llava_measurement(dev, frames_total=2048, layers=LAYERS, steps=2, warmup=1, pool_n=48):
    """BASELINE configs[4] on ONE GPU: the DPSelect + PivotKV share of a 2048-frame LLaVA-Video-Qwen2-7B prefill, through
    the same plugin surface.  DPSelect (ratio 1.0, patch_sync False) on the SigLIP patch embeddings [1, T, 729, 1152] bf16
    (3.4 GB at T = 2048); the key-patch mask [T * 729] truncated to the first T * 196 + 1 entries (the reference's quirk,
    llava_onevision.py:486); PivotKV on T / 32 chunks x 28 layers of L = 6272 pooled tokens, plain RoPE with [1, L] ids,
    YaRN factor 4, pos_embed_reforge, `dynamic_compression_ratio` with max_input_length 40000: ratio 40000 / (T * 196 + 1),
    keep = 624 at 2048 frames (configs/llava_video/retake_llava-video_*.yaml).  The LLaVA patch shifts a CLONE of the ids
    per layer (llava_onevision.py:76-88)."""
    import retake._native as nv
    import retake.longvideo_cache as lc
    import retake.visual_compression as vc
    import unit_check as uc
    from retake import _prefill

    N_SIGLIP, C_SIGLIP, N_POOLED = 729, 1152, 196
    T = frames_total
    L = 32 * N_POOLED
    n_chunks = T // 32
    g = torch.Generator(device=dev).manual_seed(4242)
    frames = torch.randn((1, T, N_SIGLIP, C_SIGLIP), generator=g, device=dev, dtype=torch.float32).bfloat16()
    pool = [tuple((1.7 * torch.randn((1, h, L, D), generator=g, device=dev, dtype=torch.float32)).bfloat16() for h in (Hq, Hkv, Hkv))
            for _ in range(min(pool_n, n_chunks * layers))]
    rotary = PlainRotary(dev)
    llm = types.SimpleNamespace(hidden_size=Hq * D, num_hidden_layers=layers, num_attention_heads=Hq, num_key_value_heads=Hkv)
    kwc = {"compression_ratio": 0.5, "compression_method": "pivotkv", "pos_embed_reforge": True,
           "dynamic_compression_ratio": True, "max_input_length": 40000}
    n_visual = T * N_POOLED + 1

    def run():
        cfg = types.SimpleNamespace(text_config=llm, longvideo_kwargs={"kvcache_compression": True,
                                                                       "kvcache_compression_kwargs": dict(kwc)})
        _prefill.apply_dynamic_compression_ratio(cfg, n_visual)          # what the model forward does (llava_onevision.py)
        out, mask = vc.memory_bank_compress_keyframe(frames, T, 3, sync=False)
        mask = mask[:n_visual]                                            # the reference's truncation, not a re-pooling
        cache = lc.build_kvcache(cfg)
        keep = max(1, int(cache.compression_ratio * L))
        call = 0
        for c in range(n_chunks):
            cache.keypatches_mask_chunk = mask[c * L:(c + 1) * L]
            cache.kvcache_compression = True
            pos = (torch.arange(L, device=dev) + 16 + c * L)[None].contiguous()
            for layer in range(layers):
                q, k, v = pool[call % len(pool)]
                call += 1
                p_l = cache.shift_temporal_ids_(pos.clone(), layer)      # LLaVA's patch shifts a clone per layer
                cache.update(k, v, layer, {"query_states": q, "position_ids": p_l, "rotary_emb": rotary})
            cache.after_forward()
        return cache, mask, keep

    for _ in range(warmup):
        run()
    ids = nv.profile_kernel_ids()
    nv.check(nv.lib.rtk_profile_reset(), "profile_reset")
    nv.check(nv.lib.rtk_profile_enable_mask((1 << ids["score_pass1"]) | (1 << ids["score_pass2"]) |
                                            (1 << ids["dpselect_dis"]) | (1 << ids["gather_frames"])), "profile_enable")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cache = None
    for _ in range(steps):
        cache = None
        cache, mask, keep = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nv.check(nv.lib.rtk_profile_enable(0), "profile_enable")
    kern = {k: {"launches": n, "avg_us": ms / n * 1e3} for k, (n, ms) in nv.profile_read().items()}
    # untimed: the last chunk's first / middle / last layer against one-unit launches (bitwise)
    c = n_chunks - 1
    lay = sorted({0, layers // 2, layers - 1})
    inputs = {l: pool[(c * layers + l) % len(pool)][:2] for l in lay}
    uc.check_batch_against_units(cache, lay, inputs, {l: mask[c * L:(c + 1) * L] for l in lay}, keep, rotary.inv_freq,
                                 A_SCALE, None)
    flops = 2.0 * Hq * L * L * D * layers
    res = {"metric": "frames/sec through DPSelect+PivotKV, LLaVA-Video geometry (BASELINE configs[4], 1 GPU share)",
           "value": T * steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "dtype": "bf16", "data": "synthetic",
           "retained_kv_tokens_per_s": n_chunks * layers * keep * steps / dt,
           "config": {"workload": f"DPSelect on [1,{T},{N_SIGLIP},{C_SIGLIP}] bf16 + PivotKV (plain RoPE, dynamic ratio "
                                  f"{cache.compression_ratio:.5f}, keep {keep}) on {n_chunks} chunks x {layers} layers, L={L}",
                      "keep": keep, "assembled_cache_tokens": int(cache.key_cache[0].shape[2])},
           "kernels_timed_region": kern,
           # pass 1 (every key) priced against the dense bf16 peak; pass 2 computes the unmasked columns only
           "score_pass1_frac_of_2.5PF": flops / (kern["score_pass1"]["avg_us"] * 1e-6) / 2.5e15,
           "key_patch_mask_rate": float(mask[: n_chunks * L].float().mean().item()),
           "dpselect_dis_GBps": (T * N_SIGLIP * C_SIGLIP * 2 + 4 * T * N_SIGLIP) / (kern["dpselect_dis"]["avg_us"] * 1e-6) / 1e9,
           "self_check": "batched score / keep_idx / new ids == one-unit launches (bitwise), last chunk, layers %s" % lay}
    del cache, frames, pool
    torch.cuda.empty_cache()
    return res


def decode_prologue_measurement(dev, tokens=100, prefix=40000):
    """Decode-time cost of the attention patch's prologue (qwen2_vl.py:68-86 + the cache's else-branch :319-321) for the
    LAYERS layers of one generated token after a compressed prefill of `prefix` cached rows per layer: the op-by-op route
    (continuity shift, rotary module, apply_multimodal_rotary_pos_emb, PivotKVCache.update - what the reference's patch
    runs, ~25 launches per layer) against the fused one (PivotKVCache.append_pre_rope: one kernel + the in-place id
    shift).  Wall time per step with the GPU drained at both ends; also for a 64-token text segment."""
    import retake.longvideo_cache as lc

    td = torch.bfloat16
    layers = LAYERS
    rot = Rotary(dev)
    out = {"layers": layers, "dtype": "bf16", "prefix_tokens_per_layer": prefix, "steps": tokens,
           "note": "wall time of the patch's prologue for all layers of one step, attention itself not included"}

    def proj(n):
        return tuple(torch.randn((1, n, h, D), device=dev, dtype=torch.float32).to(td).transpose(1, 2) for h in (Hq, Hkv, Hkv))

    for n, label in ((1, "decode_token"), (64, "text_segment_64")):
        res = {}
        for fused in (True, False):
            cache = lc.build_kvcache(make_cache_config(layers), reserve_tokens=prefix + (tokens + 8) * n + 64)
            cache.kvcache_compression = False
            for l in range(layers):   # the state a long compressed prompt leaves: `prefix` cached rows with their ids
                st = cache.reserve(l, prefix, torch.empty((1, Hkv, 1, D), dtype=td, device=dev))
                st.length = prefix
                cache._pos_reserve(st, 3, 3, prefix, dev)
                st.pos[:, :prefix] = torch.arange(prefix, device=dev)
                st.pos_len = prefix
            cache._pos_layers = layers
            qkv = [proj(n) for _ in range(4)]

            def step(t):
                pos = (torch.arange(n, device=dev) + prefix + 10000 + t * n).view(1, 1, n).repeat(3, 1, 1)
                for l in range(layers):
                    q, k, v = qkv[(t + l) % 4]
                    if fused:
                        assert cache.append_pre_rope(q, k, v, l, pos, rot, MROPE) is not None
                    else:
                        cache.shift_temporal_ids_(pos, l)
                        cos, sin = rot(v, pos)
                        qr, kr = lc.apply_multimodal_rotary_pos_emb(q, k, cos, sin, MROPE)
                        cache.update(kr, v, l, {"sin": sin, "cos": cos, "query_states": qr, "position_ids": pos,
                                                "rotary_emb": rot, "mrope_section": MROPE})

            for t in range(5):
                step(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(tokens):
                step(5 + t)
            torch.cuda.synchronize()
            res["fused_us_per_step" if fused else "op_by_op_us_per_step"] = (time.perf_counter() - t0) / tokens * 1e6
            del cache
            torch.cuda.empty_cache()
        res["speedup"] = res["op_by_op_us_per_step"] / res["fused_us_per_step"]
        out[label] = res
    return out


# ---------------------------------------------------------------------------------------------------
# The contract line.  stdout carries ONE short JSON line (< 4 KB: the driver keeps an 8 KB tail of stdout and parses the
# last line of it; a 32 KB line once left it with nothing to parse).  Everything else bench.py measures - per-kernel tables,
# the companions' full records, the memory split - is the REPORT: written to bench_report.json (repo root, and gpurun_out/
# when that directory exists) and to stderr; tools/show_bench.py reads the file.
# ---------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096
REPORT_NAME = "bench_report.json"
_COMPANIONS = ("real_geometry", "no_keypatch_mask", "llava_workload", "rotary_module_called", "reference_rounding",
               "fast_rounding", "fp16_dtype", "fp32_parity_dtype")


def _sig(x, digits=6):
    """Floats to `digits` significant digits (the report file keeps full precision)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x == 0.0 or x != x or x in (float("inf"), float("-inf")):
        return x
    return float(f"{x:.{digits}g}")


def _short(s, n=120):
    return s if len(s) <= n else s[: n - 3] + "..."


def contract_line(report: dict, report_path=None) -> dict:
    """The short line of a full report: exactly the contract keys + one number per extra measurement."""
    line = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data"):
        if k in report:
            line[k] = _sig(report[k], 8)
    cfg = report.get("config", {})
    line["config"] = {k: (_short(v) if isinstance(v, str) else _sig(v)) for k, v in cfg.items()
                      if k in ("workload", "geometry", "score_rounding", "frames", "chunks", "layers", "chunk_tokens", "keep",
                               "parallelism", "transport", "key_patch_mask_rate", "assembled_cache_tokens")}
    if report.get("roofline") is not None:
        r = report["roofline"]
        line["roofline"] = {k: _sig(r.get(k)) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                        "units_per_launch")}
        line["roofline"]["avg_launch_us"] = _sig(r.get("avg_launch_us"))
    cb = report.get("cpu_baseline")
    line["cpu_baseline"] = None if cb is None else {
        "value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": _short(cb["sample"], 128)}
    for k in ("speedup_vs_cpu_baseline", "retained_kv_tokens_per_s"):
        if k in report:
            line[k] = _sig(report[k])
    sc = report.get("self_check")
    if sc is not None:
        line["self_check"] = sc["status"] if isinstance(sc, dict) else sc
    for k in ("sharded_equals_sequential", "rccl_world_size", "p2p_world_size", "host_staged_world_size", "phase_ms"):
        if k in report:
            line[k] = report[k]
    n1 = report.get("n1_same_arithmetic")
    if isinstance(n1, dict):   # frames/s of the sharded path at world size 1: the base of the --gpus N curve
        line["n1_same_arithmetic"] = _sig(n1["value"]) if "value" in n1 else _short(n1.get("error", "failed"), 80)
    m = report.get("memory")
    if m:
        line["memory"] = {"product_peak_bytes": m["product_peak_bytes"], "peak_allocated_bytes": m["peak_allocated_bytes"],
                          "product_peak_over_reference_formula": _sig(m["product_peak_over_reference_formula"], 4)}
    hk = report.get("roofline_hbm_kernels")
    if hk:
        line["roofline_hbm_kernels"] = {k: _sig(v["frac"], 3) for k, v in hk.items()}
    comp = {k: _sig(report[k]["value"], 5) for k in _COMPANIONS if isinstance(report.get(k), dict) and "value" in report[k]}
    for g, c in (report.get("pre_rope_prologue") or {}).items():
        if isinstance(c, dict) and "value" in c:
            comp["pre_rope/" + g] = _sig(c["value"], 5)
    if "overlap" in report:
        comp["overlap_streams"] = _sig(report["overlap"]["value"], 5)
    dp = report.get("decode_prologue")
    if dp:
        comp["decode_prologue_fused_us"] = _sig(dp["decode_token"]["fused_us_per_step"], 4)
    if comp:
        line["companions_frames_per_s"] = comp
    if report_path:
        line["report"] = report_path
    # never above the limit: the optional summaries go first, the contract keys stay
    for drop in ("companions_frames_per_s", "roofline_hbm_kernels", "memory", "phase_ms"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) >= LINE_LIMIT:
        raise AssertionError("bench.py: the contract line exceeds %d bytes" % LINE_LIMIT)
    return line


def emit(report: dict, path=None):
    """Report -> `path` (default: bench_report.json at the repo root + in gpurun_out/) and stderr; the short contract line
    -> stdout, LAST."""
    text = json.dumps(report)
    where = None
    targets = [path] if path else [os.path.join(d, REPORT_NAME) for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)]
    for t in targets:
        try:
            with open(t, "w") as f:
                f.write(text + "\n")
            where = where or os.path.relpath(t, ROOT)
        except OSError:
            pass
    line = contract_line(report, where)
    sys.stderr.write(text + "\n")
    sys.stderr.flush()
    try:   # anything a C library left in stdio's buffer (RCCL's banner) goes out BEFORE the line, not at exit after it
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)
    return line


def n1_same_arithmetic(args, timeout=300):
    """frames/s of the SHARDED path at world size 1 over RCCL (RETAKE_FORCE_SHARDED=1) on the same video: the point a
    scaling curve over `bench.py --gpus N` should be held against - same host code (ShardedPivotKV: provisional ids,
    deferred rotation, offsets, assembly) and kernels as the N > 1 lines, where the plain N = 1 line runs the sequential
    cache.  Measured in a CHILD process after this process has released its tensors (RCCL never enters the process that
    prints the contract line; a child that hangs is killed after `timeout` s and reported as such)."""
    import socket
    import subprocess
    import tempfile

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    with tempfile.TemporaryDirectory() as td:
        rep = os.path.join(td, "world1.json")
        cmd = [sys.executable, os.path.abspath(__file__), "--frames", str(args.frames), "--layers", str(args.layers),
               "--steps", "2", "--warmup", "1", "--dtype", args.dtype, "--pool", str(args.pool), "--no-cpu-baseline",
               "--no-self-check", "--no-extras", "--report", rep]
        env = {**os.environ, "RETAKE_FORCE_SHARDED": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
        env.pop("RANK", None)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        except subprocess.TimeoutExpired:
            return {"error": f"the world-size-1 sharded run did not finish in {timeout} s"}
        if r.returncode != 0 or not os.path.exists(rep):
            return {"error": "the world-size-1 sharded run failed: " + r.stderr[-400:]}
        full = json.load(open(rep))
    return {k: full[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "phase_ms", "roofline", "rccl_world_size")
            if k in full}


def main():
    global OVERLAP_STREAMS, SCORE_ROUNDING
    args = parse()
    OVERLAP_STREAMS = args.streams
    SCORE_ROUNDING = args.score_rounding
    for kv in args.cache_option:
        k_, v_ = kv.split("=", 1)
        CACHE_EXTRA[k_] = {"0": False, "1": True, "false": False, "true": True}.get(v_.lower(), v_)
    set_geometry(args.geometry)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # launched without torchrun: start the ranks as child processes (nothing here has touched the GPU yet)
        import socket
        import subprocess

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world > 1 or os.environ.get("RETAKE_FORCE_SHARDED") == "1":  # the env switch runs the sharded path at N=1
        from retake import sharded

        if args.geometry != "baseline":
            raise SystemExit("bench.py --gpus N shards the BASELINE geometry (configs[3]); use --geometry baseline")
        return sharded.bench_main(args, rank, world, local_rank)

    import retake._native as nv

    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    tdtype = TORCH_DTYPE[args.dtype]
    es = 4 if args.dtype == "fp32" else 2
    T = args.frames // FRAMES_PER_ROW     # rows of the frame bank (= frames; temporal grids of 2 frames for qwen448)
    L = FRAMES_PER_CHUNK * N_PATCH
    n_chunks = T // FRAMES_PER_CHUNK
    frames = torch.cat([chunk_frames(c, dev, tdtype) for c in range(n_chunks)])[None]
    pool = [pool_set(i, dev, tdtype, projection_layout=args.pre_rope) for i in range(min(args.pool, n_chunks * args.layers))]
    pos_base = [chunk_position_ids(c, dev) for c in range(n_chunks)]
    rotary = Rotary(dev)
    masks = None
    if args.pre_rope:
        args.no_self_check = True   # the self-check replays rotated inputs through one-unit launches
        args.no_extras = True

    for _ in range(args.warmup):
        run_video(frames, pool, masks, pos_base, rotary, args.layers, tdtype, args.pre_rope)
    torch.cuda.synchronize()
    # HIP events bracket the dominant kernels (the two score passes) on their launch streams INSIDE the timed
    # region; timing every small kernel as well costs ~9 % of wall time, so the full per-kernel table comes
    # from a short extra segment after the timed region.
    use_events = not args.no_kernel_events
    if use_events:
        ids = nv.profile_kernel_ids()
        nv.check(nv.lib.rtk_profile_reset(), "profile_reset")
        nv.check(nv.lib.rtk_profile_enable_mask((1 << ids["score_pass1"]) | (1 << ids["score_pass2"])), "profile_enable")
    gc.collect()
    torch.cuda.synchronize()
    resident_inputs = torch.cuda.memory_allocated()     # frame bank + (q, k, v) pool + ids: the bench's own tensors
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    retained = 0
    cache = None
    for _ in range(args.steps):
        cache = None     # the previous video's cache is released first, as after a finished `generate`
        r, cache, kp_mask = run_video(frames, pool, masks, pos_base, rotary, args.layers, tdtype, args.pre_rope)
        retained += r
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cache.check()   # a bounded device-side wait that ran out (the in-launch id shift) raises: no number for a wrong path
    peak_alloc, peak_reserved = torch.cuda.max_memory_allocated(), torch.cuda.max_memory_reserved()
    mem = memory_block(cache, peak_alloc, peak_reserved, resident_inputs,
                       frames.numel() * frames.element_size() + 5 * T * N_PATCH, args.layers, L, n_chunks, es)
    prof, prof_all = {}, {}
    if use_events:
        nv.check(nv.lib.rtk_profile_enable(0), "profile_enable")
        prof = nv.profile_read()
    check = self_check(cache, pool, kp_mask, n_chunks, args.layers, rotary) if not args.no_self_check else None
    checksum = cache_checksum([cache.key_cache[l] for l in range(args.layers)],
                              [cache.value_cache[l] for l in range(args.layers)], cache.position_cache)
    del cache
    if use_events:
        # untimed: every kernel, up to 64 chunks' worth of frames / updates on one stream (>= 64 launches of the per-chunk
        # kernels, ~1800 of the per-update ones)
        saved = OVERLAP_STREAMS
        OVERLAP_STREAMS = 0
        nv.check(nv.lib.rtk_profile_reset(), "profile_reset")
        nv.check(nv.lib.rtk_profile_enable(1), "profile_enable")
        run_video(frames[:, : min(T, 64 * FRAMES_PER_CHUNK)], pool, masks, pos_base, rotary, args.layers, tdtype, args.pre_rope)
        torch.cuda.synchronize()
        dp_names = ("dpselect_dis", "dpselect_select", "gather_frames")
        prof_all = {k: v for k, v in nv.profile_read().items() if k not in dp_names}
        nv.check(nv.lib.rtk_profile_reset(), "profile_reset")   # DPSelect kernels at full size
        import retake.visual_compression as vc
        vc.memory_bank_compress_keyframe(frames, T, 3, sync=False)
        torch.cuda.synchronize()
        nv.check(nv.lib.rtk_profile_enable(0), "profile_enable")
        prof_all.update({k: v for k, v in nv.profile_read().items() if k in dp_names})
        OVERLAP_STREAMS = saved

    ms_per_step = dt / args.steps * 1e3
    fps = args.frames * args.steps / dt
    out = {
        "metric": "frames/sec through DPSelect+PivotKV @2048 frames; retained-KV-tokens/sec",
        "value": fps, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "retained_kv_tokens_per_s": retained / dt,
        # (kept under 120 characters: the driver's record truncates longer strings)
        "config": {"workload": ("BASELINE configs[2]" if args.geometry == "baseline" else GEOMETRIES[args.geometry][6])
                               + f": {args.frames}-frame video, DPSelect [1,{T},{N_PATCH},{C_EMB}] + PivotKV 4x, "
                                 f"{n_chunks} chunks x {args.layers} layers, L={L}",
                   "workload_detail": f"Qwen2-VL-7B head geometry Hq={Hq}, Hkv={Hkv}, D={D}; DPSelect async at ratio 1.0 (shipped "
                                      f"default); PivotKV with pos_embed_reforge + M-RoPE {MROPE}, YaRN attention_scaling",
                   "geometry": args.geometry, "score_rounding": args.score_rounding,
                   "cache_kwargs": cache_kwargs(),
                   "native_rope": "product default: on for inv_freq rotary modules (no config key set)",
                   "update_call": ("PivotKVCache.update_pre_rope (pre-RoPE projections, projection layout)" if args.pre_rope else
                                   "PivotKVCache.update (rotated q / k, the reference's cache_kwargs protocol) after "
                                   "shift_temporal_ids_, as the attention patch calls them"),
                   "frames": args.frames, "chunks": n_chunks, "layers": args.layers, "chunk_tokens": L, "keep": int(RATIO * L),
                   "input_pool_sets": len(pool), "worker_streams": args.streams, "parallelism": "1 GPU"},
    }
    # pass 2 computes one column mass per UNMASKED key (the reference overwrites the masked tokens' scores with 1.0,
    # longvideo_cache.py:272-274): state how many columns that is for this video's DPSelect mask
    out["config"]["key_patch_mask_rate"] = float(kp_mask.float().mean().item())
    out["config"]["pass2_live_key_fraction"] = 1.0 - out["config"]["key_patch_mask_rate"]
    if check is not None:
        out["self_check"] = check
    out["cache_checksum"] = checksum   # the sharded runs (--gpus N) print the same fingerprint of the assembled cache
    out["memory"] = mem
    if prof:
        kern = {k: {"launches": n, "avg_us": ms / n * 1e3, "total_ms": ms} for k, (n, ms) in prof_all.items()}
        out["kernels_untimed_single_stream"] = dict(kern)
        timed = {k: {"launches": n, "avg_us": ms / n * 1e3, "total_ms": ms} for k, (n, ms) in prof.items()}
        out["kernels_timed_region"] = timed
        for k, v in timed.items():
            kern[k] = v
        out["roofline"] = score_roofline(kern, args.dtype, L, T, n_chunks * args.layers * args.steps)
        # HBM-bound kernels of the path, same convention (bytes the launch has to move / avg duration).
        # The eviction scan (16.3 MB algorithmic per (layer, chunk)) is three kernels here:
        #   append          per update: K,V rows read + written to the cache tail
        #   evict_batched   per chunk : kept K~,V rows + ids + new-position tables read, kept rows + ids written,
        #                               for all `layers` units of the chunk in one launch
        #   commit_batched  per chunk : staged V rows read + written, all units in one launch
        # or, by default (in_place_compaction), ONE launch per chunk for the last two:
        #   compact_units   per chunk : kept K~ rows read + written re-rotated, kept V rows read + written inside the
        #                               tail, kept-index + ids read, ids written - nothing staged
        keep = int(RATIO * L)
        ap_bytes = 2 * 2 * Hkv * L * D * es
        prep_bytes = (2 * Hq + 5 * Hkv) * L * D * es        # q: read + q~; k: read + k~ + tail; v: read + tail
        # kept K and V rows read + written, kept-index read, ids read + written; the fp32 cos/sin tables only when a
        # separate launch wrote them (third-party rotary module) - with the native RoPE the kernel computes them
        tables = "rope_table" in kern
        # eviction launch: kept K rows read (k~) + written (re-rotated, straight into the cache), the V rows whose source
        # lies inside the destination range read + parked (fraction f of the kept rows, measured by the self-check;
        # ~ratio), kept-index read, ids read + written.  Placement launch: every kept V row read (in place or parked)
        # + written to the head of the tail.
        f_low = (check or {}).get("staged_fraction", RATIO)
        evu_bytes = (2 + 2 * f_low) * Hkv * keep * D * es + (2 * keep * D * 4 if tables else 0) + 8 * keep + 2 * 8 * 3 * keep
        cmu_bytes = 2 * Hkv * keep * D * es + 8 * keep
        cpu_bytes = 4 * Hkv * keep * D * es + 8 * keep + 2 * 8 * 3 * keep
        ev_bytes = 5 * L + 2 * Hkv * L * D * es + 2 * Hkv * keep * D * es + 8 * 3 * (L + keep)
        dp_bytes = T * N_PATCH * C_EMB * es + 4 * T * N_PATCH
        ga_bytes = 2 * T * N_PATCH * C_EMB * es
        extra = {}
        # what a plain device copy reaches on THIS GPU (measured now, outside the timed region): every HBM fraction below is
        # quoted against the nominal 8 TB/s (`frac`) and against this (`frac_of_measured_copy`)
        achievable = hbm_achievable(frames.device)
        best_copy = achievable["copy_GBps"]
        fused = "append" not in kern   # native-RoPE path: tables + un-rotate + append are ONE kernel (unrotate_pack)
        # the attention prologue (--pre-rope): q read + rotated q written (the queries are scored where they lie: no packed
        # copy), k read + k~ + rotated tail, v read + tail
        pro_bytes = (2 * Hq + 5 * Hkv) * L * D * es
        for name, key, b in (("append", "append", ap_bytes),
                             ("prologue", "prologue", pro_bytes),
                             ("prepare_fused" if fused else "unrotate", "unrotate_pack",
                              prep_bytes if fused else 2 * (Hq + Hkv) * L * D * es),
                             ("evict_batched", "evict_batched", evu_bytes * args.layers),
                             ("commit_batched", "commit_batched", cmu_bytes * args.layers),
                             ("compact_units", "compact_units", cpu_bytes * args.layers),
                             ("dpselect_dis", "dpselect_dis", dp_bytes), ("gather_frames", "gather_frames", ga_bytes)):
            if key in kern:
                gbs = b / (kern[key]["avg_us"] * 1e-6) / 1e9
                tr = None
                if args.dtype == "bf16" and T == 2048 and args.layers == LAYERS and args.geometry == "baseline":
                    tr = pmc_traffic({"append": "append_kernel", "evict_batched": "evict_batched_kernel",
                                      "commit_batched": "place_batched_kernel", "compact_units": "compact_units_kernel",
                                      "prepare_fused": "prepare_native_kernel"}.get(name, "\0"))[0]
                extra[name] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": gbs / HBM_PEAK_GBS, "frac_of_measured_copy": gbs / best_copy, "traffic": tr,
                               "algorithmic_bytes_per_launch": b}
        one_launch = "compact_units" in kern and kern["compact_units"]["launches"] >= kern.get("evict_batched", {"launches": 0})["launches"]
        if (one_launch or all(k in kern for k in ("evict_batched", "commit_batched"))) and ("append" in kern or fused):
            # SURVEY §8(d) "PivotKV eviction scan (P6-P13)": mask override + select + kept-row gather / re-rotation +
            # id bookkeeping + compaction of one (layer, chunk) unit, against SURVEY's algorithmic byte count.  Every
            # stage is one launch per chunk covering all layers, so a unit's share is 1/layers of each launch.
            per_chunk = {"pivotkv_select": 1, "pivotkv_emit": 1, "rope_table": 1}
            per_chunk.update({"compact_units": 1} if one_launch else {"evict_batched": 1, "commit_batched": 1})
            if one_launch:
                evu_bytes, cmu_bytes = cpu_bytes, 0
            stages = {k: kern[k]["avg_us"] * n / args.layers for k, n in per_chunk.items() if k in kern}
            t_scan = sum(stages.values()) * 1e-6
            t_data = (stages.get("evict_batched", 0) + stages.get("commit_batched", 0) + stages.get("compact_units", 0)) * 1e-6
            moved = evu_bytes + cmu_bytes
            # The scan is INDEX-DRIVEN: the selection reads 5 bytes per token, the data launch(es) touch kept rows only.  Its
            # algorithmic bytes are therefore the bytes those rows have to move (`moved`), NOT SURVEY 8(d)'s 16.3 MB per unit,
            # which charges a read of every K / V row of the chunk: by that count the compaction launch alone would run at
            # ~1.4x the HBM peak.  The SURVEY figure is kept below as an equivalent only.
            extra["eviction_scan_per_unit"] = {
                "bound": "hbm", "achieved": moved / t_scan / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": moved / t_scan / 1e9 / HBM_PEAK_GBS, "frac_of_measured_copy": moved / t_scan / 1e9 / best_copy,
                "traffic": None, "algorithmic_bytes_per_unit": moved, "us_per_unit": t_scan * 1e6,
                "stages_us_per_unit": stages,
                "definition": "select + data launch(es) of one (layer, chunk) unit; bytes = kept K rows read + written, kept V "
                              "rows read + written, kept-index + ids read, ids written",
                "data_launches_only": {"achieved": moved / t_data / 1e9, "frac": moved / t_data / 1e9 / HBM_PEAK_GBS,
                                       "frac_of_measured_copy": moved / t_data / 1e9 / best_copy, "us_per_unit": t_data * 1e6},
                "survey_bytes_equivalent": {"bytes_per_unit": ev_bytes, "GBps": ev_bytes / t_scan / 1e9,
                                            "over_peak": ev_bytes / t_scan / 1e9 / HBM_PEAK_GBS,
                                            "data_launches_only_over_peak": ev_bytes / t_data / 1e9 / HBM_PEAK_GBS,
                                            "note": "SURVEY 8(d)'s byte count (reads every K / V row): not a rate any kernel of an "
                                                    "index-driven in-place compaction achieves - do not read it as a roofline fraction"}}
            # the same plus the tail append update() owes the layer's attention (reference :238, P1): its own kernel,
            # or its byte share of the fused prepare kernel
            if "prologue" in kern:
                t_app = kern["prologue"]["avg_us"] * ap_bytes / pro_bytes
            else:
                t_app = kern["append"]["avg_us"] if not fused else kern["unrotate_pack"]["avg_us"] * ap_bytes / prep_bytes
            t_unit = t_scan + t_app * 1e-6
            moved_u = ap_bytes + moved
            extra["cache_update_per_unit"] = {
                "bound": "hbm", "achieved": moved_u / t_unit / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": moved_u / t_unit / 1e9 / HBM_PEAK_GBS, "frac_of_measured_copy": moved_u / t_unit / 1e9 / best_copy,
                "traffic": None, "algorithmic_bytes_per_unit": moved_u, "us_per_unit": t_unit * 1e6,
                "definition": "eviction scan + the unit's share of the append (K, V rows read + written to the tail)",
                "survey_bytes_equivalent": {"bytes_per_unit": ev_bytes, "GBps": ev_bytes / t_unit / 1e9,
                                            "over_peak": ev_bytes / t_unit / 1e9 / HBM_PEAK_GBS}}
        out["roofline_hbm_kernels"] = extra
        out["hbm_achievable"] = achievable
    if args.also_streams > 0 and args.streams == 0:
        # same workload with scoring / selection / eviction on worker HIP streams (PivotKVCache
        # overlap_streams): kernels of independent updates overlap, so per-kernel event durations no longer
        # measure a kernel that owns the chip; the sustained MFMA rate is total score flops / wall time
        OVERLAP_STREAMS = args.also_streams
        run_video(frames, pool, masks, pos_base, rotary, args.layers, tdtype)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            run_video(frames, pool, masks, pos_base, rotary, args.layers, tdtype)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        OVERLAP_STREAMS = 0
        flops_step = 2.0 * 2.0 * Hq * L * L * D * n_chunks * args.layers   # two contractions per update
        out["overlap"] = {"worker_streams": args.also_streams, "value": args.frames * args.steps / dt2, "unit": "frames/s",
                          "ms_per_step": dt2 / args.steps * 1e3,
                          "sustained_score_tflops": flops_step * args.steps / dt2 / 1e12}
    if not args.no_extras and args.geometry == "baseline" and args.dtype == "bf16" and args.score_rounding == "fp32":
        # Companions of the headline, each measured the same way in small AFTER the timed region above (none of them
        # enters `value`): (1) the real Qwen2-VL geometry SURVEY 8(d) names beside the synthetic one; (2) the price
        # of reproducing the reference's own bf16 roundings bit for bit; (3) the parity dtype.
        del frames, pool
        torch.cuda.empty_cache()
        out["real_geometry"] = companion_measurement(dev, args.frames, args.layers, "bf16", max(2, args.steps), 1, args.pool,
                                                     geometry="qwen448", time_all_kernels=True)
        out["real_geometry"]["config"]["workload"] = (
            "real Qwen2-VL-7B geometry at 448 px 16:9: 1024 temporal grids x 144 merged tokens x 3584 channels, "
            "chunk = 16 grids = 2304 tokens (cal_flops.py:8,47; qwen2_vl.py:477-491)")
        # the reference's videomme config runs WITHOUT visual compression (configs/qwen2_vl/retake_qwen2-vl_videomme.yaml:6-17):
        # no DPSelect, no key-patch mask - pass 2 then computes every column (the headline's mask rate is what DPSelect
        # produces on i.i.d. frames and saves pass 2 that share of its work)
        out["no_keypatch_mask"] = companion_measurement(dev, args.frames, args.layers, "bf16", 2, 1, args.pool, no_visual=True)
        out["no_keypatch_mask"]["note"] = ("visual_compression off (the reference's videomme config): PivotKV only, no key-patch "
                                           "mask, score pass 2 over all keys")
        # BASELINE configs[4], its single-GPU share: LLaVA-Video geometry (SigLIP patches, plain RoPE, dynamic ratio 0.0996)
        out["llava_workload"] = llava_measurement(dev, args.frames, args.layers, 2, 1, args.pool)
        # the attention patch's fused prologue: the same step fed with PRE-RoPE projections in the projection layout
        # (what the patched HF attention hands over on the GPU), at both geometries; default operands (the reference's
        # round-tripped q~ / k~) and the opt-in pre-RoPE operands (queries scored where they lie)
        out["pre_rope_prologue"] = {
            "note": "PivotKVCache.update_pre_rope: continuity shift + rotary tables + RoPE of q / k + cache append + "
                    "scoring operands in ONE kernel per update (replaces position_shift + HF's eager RoPE ops + "
                    "prepare); the rotated queries go to a scratch tensor of the projection layout",
            "real_geometry": companion_measurement(dev, args.frames, args.layers, "bf16", 2, 1, args.pool,
                                                   geometry="qwen448", pre_rope=True, time_all_kernels=True),
            "baseline_geometry": companion_measurement(dev, args.frames, args.layers, "bf16", 2, 1, args.pool,
                                                       pre_rope=True, time_all_kernels=True),
            "real_geometry_pre_rope_operands": companion_measurement(
                dev, args.frames, args.layers, "bf16", 2, 1, args.pool, geometry="qwen448", pre_rope=True,
                cache_extra={"prologue_operands": "pre_rope"}, time_all_kernels=True)}
        # the bit-faithful opt-out of the native RoPE: the rotary module is CALLED for the tables (reference :249, :298)
        out["rotary_module_called"] = companion_measurement(dev, args.frames, args.layers, "bf16", 2, 1, args.pool,
                                                            cache_extra={"native_rope": False})
        out["rotary_module_called"]["note"] = "native_rope: False - the opt-out; tables from the rotary module + merge kernel"
        out["reference_rounding"] = companion_measurement(dev, args.frames, args.layers, "bf16", 2, 1, args.pool,
                                                          score_rounding="reference")
        out["reference_rounding"]["note"] = ("score_rounding='reference': the reference's bf16 logits / probabilities / sums "
                                             "(longvideo_cache.py:264-270) reproduced rounding by rounding")
        out["fast_rounding"] = companion_measurement(dev, args.frames, args.layers, "bf16", 2, 1, args.pool,
                                                     score_rounding="fast")
        out["fast_rounding"]["note"] = ("score_rounding='fast' (opt in): q~ pre-scaled and both operands as fp16 on "
                                        "v_mfma_f32_32x32x16_f16, two instructions per logit; scores within ~1e-5 of the default")
        out["fp16_dtype"] = companion_measurement(dev, args.frames, args.layers, "fp16", 2, 1, args.pool)
        out["fp16_dtype"]["note"] = "float16 tensors (RTK_F16): fp16 rounding chains, exact fp16 products on the fp16 matrix instruction"
        out["fp32_parity_dtype"] = companion_measurement(dev, args.frames, args.layers, "fp32", 1, 1, args.pool,
                                                         warmup_chunks=2)
        # decode / text prefill: the same patch's prologue for segments that are not compressed
        out["decode_prologue"] = decode_prologue_measurement(dev)
        # the base of the scaling curve: the SHARDED path (`bench.py --gpus N`'s host code and kernels) at world size 1
        out["n1_same_arithmetic"] = n1_same_arithmetic(args)
        frames = torch.cat([chunk_frames(c, dev, tdtype) for c in range(min(n_chunks, 4))])[None]
    if not args.no_cpu_baseline:
        sample_T = 128
        out["cpu_baseline"] = cpu_baseline(args, frames[:, :sample_T].float().cpu().numpy(), args.cpu_sample_updates)
        out["speedup_vs_cpu_baseline"] = fps / out["cpu_baseline"]["value"]
    emit(out, args.report)


if __name__ == "__main__":
    main()
