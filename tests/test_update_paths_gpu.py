"""The one-call entry points (rtk_pivotkv_update / rtk_pivotkv_flush, ABI 13) and the attention prologue
(PivotKVCache.update_pre_rope) against the stage-by-stage route, torch restatements and the CPU oracle.

Reference rows: longvideo_cache.py:217-323 (update), qwen2_vl.py:55-86 / llava_onevision.py:59-141 (attention prologue).
"""
import ctypes as C
import types

import numpy as np
import pytest
import torch

import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

Hq, Hkv, D = 28, 4, 128
SEC = [16, 24, 24]
A = synth.YARN_FACTOR4_ATTENTION_SCALING


def dev():
    return torch.device("cuda:0")


def cfg(layers, ratio=0.25, reforge=True, **extra):
    kw = {"compression_ratio": ratio, "compression_method": "pivotkv", "pos_embed_reforge": reforge}
    kw.update(extra)
    return types.SimpleNamespace(hidden_size=Hq * D, num_hidden_layers=layers, num_attention_heads=Hq,
                                 num_key_value_heads=Hkv,
                                 longvideo_kwargs={"kvcache_compression": True, "kvcache_compression_kwargs": kw})


def rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def native_tables(pos, rot, dtype, mrope=True):
    """[L, D] cos / sin of `pos` ([3, 1, L] or [1, L]) exactly as the kernels compute them (rtk_rope_table), rounded to
    `dtype` like a rotary module's `.to(x.dtype)`."""
    import retake._native as nv

    P, L = (3, pos.shape[-1]) if pos.ndim == 3 else (1, pos.shape[-1])
    p2 = pos.reshape(P, L).contiguous()
    cos = torch.empty((L, D), dtype=torch.float32, device=dev())
    sin = torch.empty_like(cos)
    sec = (C.c_int * 3)(*SEC) if P == 3 else None
    nv.check(nv.lib.rtk_rope_table(nv.ptr(p2), L, P, L, nv.ptr(rot.inv_freq), D, rot.attention_scaling, sec,
                                   3 if P == 3 else 0, nv.round_mode(dtype), nv.ptr(cos), nv.ptr(sin), nv.stream()),
             "rtk_rope_table")
    return cos.to(dtype), sin.to(dtype)


def projections(seed, L, dtype):
    """q0, k0, v0 as q_proj / k_proj / v_proj hand them over: [1, L, H*D] memory, transposed [1, H, L, D] views."""
    g = torch.Generator(device=dev()).manual_seed(seed)
    return tuple((1.7 * torch.randn((1, L, h, D), generator=g, device=dev())).to(dtype).transpose(1, 2) for h in (Hq, Hkv, Hkv))


def chunk_ids(c, L, mrope=True):
    if mrope:
        return torch.from_numpy(synth.mrope_position_ids(10 + 7 * c, L // 64, 8, 8, hw0=2)).to(dev())
    return (torch.arange(L, device=dev()) + 100 + 9 * c * L).view(1, L)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("ratio", [0.25, 1])
def test_one_call_update_and_flush_equal_the_staged_route(dtype, ratio):
    """3 chunks x 3 layers at L = 640 (over the L >= 512 gate of the deferred selection): the cache driven through
    rtk_pivotkv_update / rtk_pivotkv_flush holds bitwise what the stage-by-stage route leaves - keys, values, ids,
    evicted-token counts, scores and kept indices of the last unit."""
    import retake.longvideo_cache as lc

    layers, n_chunks, L = 3, 3, 640
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())

    def run(cache):
        calls = 0
        for c in range(n_chunks):
            pos = chunk_ids(c, L)
            cache.keypatches_mask_chunk = torch.from_numpy(np.random.default_rng(c).uniform(size=L) < 0.3).to(dev())
            cache.kvcache_compression = True
            for l in range(layers):
                q0, k0, v = synth.qkv_chunk(700 + 10 * c + l, Hq, Hkv, L, D)
                cache.shift_temporal_ids_(pos, l)
                q = synth.rope_forward(torch.from_numpy(q0).to(dev()), pos, rot, SEC).to(dtype)
                k = synth.rope_forward(torch.from_numpy(k0).to(dev()), pos, rot, SEC).to(dtype)
                vt = torch.from_numpy(v).to(dev()).to(dtype)
                kw = {"query_states": q, "position_ids": pos, "rotary_emb": rot, "mrope_section": SEC, "sin": None}
                ko, vo = cache.update(k, vt, l, kw)
                assert set(kw) == {"sin"}       # the reference's pops (:235, :241-243)
                assert torch.equal(ko[:, :, -L:], k) and torch.equal(vo[:, :, -L:], vt)
                calls += cache._batch.c_pending
            if c == n_chunks - 1:
                last = (cache.last_scores, cache.last_keep_indices.clone())
            cache.after_forward()
        return calls, last

    one, staged = lc.build_kvcache(cfg(layers, ratio)), lc.build_kvcache(cfg(layers, ratio, one_call_update=False))
    n_one, last_one = run(one)
    n_staged, last_staged = run(staged)
    assert n_staged == 0 and n_one > 0, "the one-call route was not taken"
    keep = max(1, int(ratio * L))
    for l in range(layers):
        assert one.key_cache[l].shape == (1, Hkv, n_chunks * keep, D)
        assert torch.equal(one.key_cache[l], staged.key_cache[l])
        assert torch.equal(one.value_cache[l], staged.value_cache[l])
        assert torch.equal(one.position_cache[l], staged.position_cache[l])
    assert one.num_evicted_tokens == staged.num_evicted_tokens == [n_chunks * (L - keep)] * layers
    if ratio != 1:
        assert torch.equal(last_one[0], last_staged[0])
    assert torch.equal(last_one[1], last_staged[1])


@pytest.mark.parametrize("operands", ["reference", "pre_rope"])
@pytest.mark.parametrize("mrope", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
def test_prologue_outputs_bitwise(dtype, mrope, operands):
    """update_pre_rope on two chunks x two layers: the rotated queries, the cache tail, the ids and the scoring operands
    are bit for bit what torch computes op by op (one rounding per op, like the reference's eager chain) from the
    kernels' own tables: q_rot = (q0*cos) + (rotate_half(q0)*sin) over the shifted ids; tail K the same of k0, tail V = v0;
    the caller's ids are shifted in place by the flush (Qwen2-VL) or left alone (LLaVA).  Scoring operands:
      prologue_operands="reference" (default)  q~, k~ = ((x*cos) - (rotate_half(x)*sin)) / a**2 of the ROTATED rows, the
                                               un-rotation the reference applies (longvideo_cache.py:76-78) in the model dtype;
      prologue_operands="pre_rope"             q~ = q0 and k~ = k0 (queries scored where they lie when q_rot goes elsewhere)."""
    import retake.longvideo_cache as lc

    layers, L = 2, 640
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    cache = lc.build_kvcache(cfg(layers, prologue_operands=operands))
    assert lc.build_kvcache(cfg(layers)).prologue_operands == "reference"
    sec = SEC if mrope else None
    es = 4 if dtype == torch.float32 else 2
    for c in range(2):
        pos = chunk_ids(c, L, mrope)
        pos_in = pos.clone()
        cache.kvcache_compression = True
        cache.keypatches_mask_chunk = None
        for l in range(layers):
            q0, k0, v0 = projections(40 + 10 * c + l, L, dtype)
            q_keep, k_keep = q0.clone(), k0.clone()
            prev = cache.get_prev_temporal_idx(l)
            prev = int(prev) if not isinstance(prev, int) else prev
            P0 = cache.get_seq_length(l)
            # layer 0: rotation in place (q~ is then a packed copy in the score workspace); layer 1: the library's pick - for
            # 16-bit tensors a fresh tensor for the rotated queries, q0 itself scored where it lies (no copy)
            out = cache.update_pre_rope(q0, k0, v0, l, pos, rot, sec, shift_ids_in_place=mrope,
                                        query_out=q0 if l == 0 else None)
            assert out is not None, "the prologue declined a plain video chunk"
            q_rot, K, V = out
            in_place_scoring = l == 1 and dtype != torch.float32 and operands == "pre_rope"
            assert (q_rot.data_ptr() != q0.data_ptr()) == in_place_scoring and K.shape[2] == P0 + L
            assert not in_place_scoring or (torch.equal(q0, q_keep) and cache._batch.q_keep[l] is q0)
            want_ids = pos_in.clone()
            if mrope:
                want_ids[0, 0] += prev + 1 - want_ids[0, 0, 0]
            else:
                want_ids[0] += prev + 1 - want_ids[0, 0]
            cos, sin = native_tables(want_ids, rot, dtype)
            want_q = (q_keep * cos) + (rot_half(q_keep) * sin)
            want_k = (k_keep * cos) + (rot_half(k_keep) * sin)
            assert torch.equal(q_rot, want_q), f"q_rot differs in {(q_rot != want_q).sum().item()} entries"
            assert torch.equal(K[:, :, P0:], want_k) and torch.equal(V[:, :, P0:], v0)
            b = cache._batch
            assert torch.equal(b.pos_old[l].reshape(want_ids.shape), want_ids)
            if operands == "reference":   # torch's own un-rotation of the rotated rows, op by op in the tensor dtype
                # (the division on the CPU, the parity platform: ATen's device kernel multiplies by the reciprocal of a
                # scalar divisor instead, which rounds differently in fp32 - the fixtures carry the CPU's true division)
                want_qt = (((want_q * cos) - (rot_half(want_q) * sin)).cpu() / A ** 2).to(dev())
                want_kt = (((want_k * cos) - (rot_half(want_k) * sin)).cpu() / A ** 2).to(dev())
            else:
                want_qt, want_kt = q_keep, k_keep
            assert torch.equal(b.k_unrot[l], want_kt[0])
            if not in_place_scoring:
                ws = b.score_ws[(b.score_ws_base - b.score_ws.data_ptr()) + l * b.ws_stride:][: Hq * L * D * es]
                assert torch.equal(ws.view(dtype).view(Hq, L, D), want_qt[0])
            assert torch.equal(pos, pos_in), "the caller's ids move at the flush, not before"
        last_prev = cache.get_prev_temporal_idx(layers - 1)   # (flushes: read it the way the reference would, then undo)
        cache.after_forward()
        if mrope:   # the flush applied the LAST layer's shift, what the reference's layer loop leaves in the tensor
            assert pos[0, 0, 0].item() >= 0 and torch.equal(pos[1:], pos_in[1:])
            assert torch.equal(pos[0, 0] - pos[0, 0, 0], pos_in[0, 0] - pos_in[0, 0, 0])
        else:
            assert torch.equal(pos, pos_in)
        del last_prev


@pytest.mark.parametrize("L", [640, 2304])
def test_prologue_cache_against_oracle_fp32(L):
    """fp32, M-RoPE, reforge, two chunks: the cache the prologue route builds against the CPU oracle fed with the ROTATED
    tensors the reference's attention patch would have produced - kept indices bit-exact on tie-free boundaries, kept K
    within 1e-5, kept V and ids exact (north_star's bar; SURVEY A8: pre-RoPE operands are allowed inside it)."""
    import retake.longvideo_cache as lc

    ratio = 0.25
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    rot_cpu = synth.RotaryStub(synth.inv_freq(D), A)
    cache = lc.build_kvcache(cfg(1, ratio))
    oc = orc.OraclePivotKV(Hq, Hkv, D, ratio, True)
    keep = int(ratio * L)
    fragile = 0
    for c in range(2):
        q0, k0, v = synth.qkv_chunk(900 + c, Hq, Hkv, L, D)
        pos = torch.from_numpy(synth.mrope_position_ids(5 + 40 * c, L // 64, 8, 8, hw0=2))
        prev = oc.get_prev_temporal_idx(0)
        pos_sh = pos.clone()
        pos_sh[0, 0] += prev + 1 - pos_sh[0, 0, 0]
        q = synth.rope_forward(torch.from_numpy(q0), pos_sh, rot_cpu, SEC)
        k = synth.rope_forward(torch.from_numpy(k0), pos_sh, rot_cpu, SEC)
        oc.update(k.numpy(), v, 0, q=q.numpy(), position_ids=pos_sh.numpy(), rotary=rot_cpu, mrope_section=SEC)
        qd = torch.from_numpy(q0).to(dev()).transpose(1, 2).contiguous().transpose(1, 2)
        kd = torch.from_numpy(k0).to(dev()).transpose(1, 2).contiguous().transpose(1, 2)
        cache.kvcache_compression = True
        out = cache.update_pre_rope(qd, kd, torch.from_numpy(v).to(dev()), 0, pos.to(dev()), rot, SEC)
        assert out is not None
        score = cache.last_scores.cpu().numpy()
        idx = cache.last_keep_indices.cpu().numpy()
        cache.after_forward()
        so = oc.last["score"]
        assert np.abs(score - so).max() < 5e-6
        srt = np.sort(so)[::-1]
        if srt[keep - 1] - srt[keep] > 2e-5:
            assert np.array_equal(idx, oc.last["keep_idx"])
            assert np.abs(cache.key_cache[0][:, :, -keep:].cpu().numpy() - oc.last["kept_k"]).max() <= 1e-5
            assert np.array_equal(cache.value_cache[0][:, :, -keep:].cpu().numpy(), oc.last["kept_v"])
            assert np.array_equal(cache.position_cache[0][..., -keep:].cpu().numpy(), oc.last["pos"])
        else:
            fragile += 1
    assert fragile < 2, "both chunks had a fragile k-th boundary: pick another seed"


@pytest.mark.parametrize("operands", ["reference", "pre_rope"])
def test_prologue_keep_all_leaves_the_appended_rows(operands):
    """compression_ratio 1 through the prologue: nothing is scored or staged - the cache holds the values the append
    wrote and the (shifted) ids.  Keys: with pre-RoPE operands the rotated rows the append wrote stay (they differ from the
    staged route's only by the rounding of the reference's un-rotate / re-rotate round trip); with the reference's
    operands (default) the kept keys ARE that round trip - equal to the staged route's up to the rare table entry where the
    rotary module's sin / cos and the kernel's correctly rounded ones land on different bf16 values."""
    import retake.longvideo_cache as lc

    layers, L, dtype = 2, 640, torch.bfloat16
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    cache = lc.build_kvcache(cfg(layers, 1, prologue_operands=operands))
    eager = lc.build_kvcache(cfg(layers, 1, one_call_update=False))
    for c in range(2):
        pos, pos_e = chunk_ids(c, L), chunk_ids(c, L)
        for l in range(layers):
            q0, k0, v0 = projections(60 + 10 * c + l, L, dtype)
            qe, ke = q0.clone(), k0.clone()
            cache.kvcache_compression = eager.kvcache_compression = True
            assert cache.update_pre_rope(q0, k0, v0, l, pos, rot, SEC) is not None
            assert cache._batch.keep_all and cache._batch.partials is None, "a keep-all batch allocated scoring scratch"
            eager.shift_temporal_ids_(pos_e, l)
            cos, sin = rot(v0, pos_e)
            qr, kr = lc.apply_multimodal_rotary_pos_emb(qe, ke, cos, sin, SEC)
            eager.update(kr, v0, l, {"query_states": qr, "position_ids": pos_e, "rotary_emb": rot, "mrope_section": SEC})
            assert torch.equal(q0, qr) or (q0 != qr).float().mean().item() < 1e-4   # module vs kernel tables: <= 1 ulp, rare
        cache.after_forward()
        eager.after_forward()
        assert torch.equal(pos, pos_e)
    for l in range(layers):
        assert torch.equal(cache.value_cache[l], eager.value_cache[l])
        assert torch.equal(cache.position_cache[l], eager.position_cache[l])
        a, b = cache.key_cache[l].float(), eager.key_cache[l].float()
        assert a.shape == b.shape == (1, Hkv, 2 * L, D)
        # the round trip mixes a channel with its rotation partner: its bf16 roundings scale with the pair's norm
        pair = (b[..., : D // 2] ** 2 + b[..., D // 2:] ** 2).sqrt()
        assert ((a - b).abs() <= 2.0 ** -5 * torch.cat((pair, pair), -1).clamp_min(1e-3)).all()
        if operands == "reference":
            assert (a != b).float().mean().item() < 1e-3, (a != b).float().mean().item()
    assert cache.last_scores is None


def test_prologue_declines_what_it_cannot_serve():
    """None - and nothing touched - for text segments (compression off), small chunks, rotary modules that have to be
    called, CPU tensors and score_rounding='reference' on pre-RoPE operands (the reference's rounding chain scores the
    reference's operands: with prologue_operands='reference' it is served, see the bf16 fixture test)."""
    import retake.longvideo_cache as lc

    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())

    class Opaque:   # no inv_freq: has to be called
        attention_scaling = 1.0

        def __call__(self, x, ids):
            return rot(x, ids)

    L = 640
    pos = chunk_ids(0, L)
    for kw, rotary, n in (({}, rot, 128), ({}, Opaque(), L),
                          ({"score_rounding": "reference", "prologue_operands": "pre_rope"}, rot, L),
                          ({"native_rope": False}, rot, L)):
        cache = lc.build_kvcache(cfg(1, **kw))
        q0, k0, v0 = projections(1, n, torch.bfloat16)
        qk = q0.clone()
        p = chunk_ids(0, n)
        assert cache.update_pre_rope(q0, k0, v0, 0, p, rotary, SEC) is None
        assert torch.equal(q0, qk) and cache.get_seq_length(0) == 0
    cache = lc.build_kvcache(cfg(1))
    cache.kvcache_compression = False
    q0, k0, v0 = projections(1, L, torch.bfloat16)
    assert cache.update_pre_rope(q0, k0, v0, 0, pos, rot, SEC) is None


def test_qwen_attention_patch_takes_the_prologue():
    """The patched Qwen2-VL SDPA attention on a stand-in module (projections + rotary), two chunks of 640 tokens and a
    reforging PivotKV cache, fp32 (the parity dtype: kept sets are decided by margins far above the arithmetic's noise):
    the fused prologue route and the op-by-op route (one_call_update off) give the same attention output and the same
    cache - ids and values exactly, keys to the 1e-5 bar."""
    import retake.longvideo_cache as lc
    import retake.qwen2_vl as rq

    hidden, L = Hq * D, 640
    torch.manual_seed(0)

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = torch.nn.Linear(hidden, Hq * D, bias=True)
            self.k_proj = torch.nn.Linear(hidden, Hkv * D, bias=True)
            self.v_proj = torch.nn.Linear(hidden, Hkv * D, bias=True)
            self.o_proj = torch.nn.Linear(Hq * D, hidden, bias=False)
            self.num_heads, self.num_key_value_heads, self.head_dim = Hq, Hkv, D
            self.num_key_value_groups, self.hidden_size = Hq // Hkv, hidden
            self.rope_scaling = {"mrope_section": SEC}
            self.attention_dropout, self.layer_idx, self.is_causal = 0.0, 0, True
            self.rotary_emb = synth.RotaryStub(synth.inv_freq(D), A, device=dev())

    attn = Attn().to(dev())
    caches = [lc.build_kvcache(cfg(1)), lc.build_kvcache(cfg(1, one_call_update=False))]
    outs = [[], []]
    with torch.no_grad():
        for c in range(2):
            x = torch.randn((1, L, hidden), device=dev()) * 0.5
            for i, cache in enumerate(caches):
                cache.kvcache_compression = True
                cache.before_forward()
                y, _, _ = rq.retake_Qwen2VLSdpaAttention_forward(attn, x, attention_mask=None, position_ids=chunk_ids(c, L),
                                                                 past_key_value=cache)
                cache.after_forward()
                outs[i].append(y.float())
    assert caches[0]._batch.c_pending == 0 and caches[0]._layers[0].length == 2 * (L // 4)
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item()
    assert torch.equal(caches[0].position_cache[0], caches[1].position_cache[0])
    assert torch.equal(caches[0].value_cache[0], caches[1].value_cache[0])
    assert (caches[0].key_cache[0] - caches[1].key_cache[0]).abs().max().item() <= 1e-5


def test_layer_state_block_tracks_the_store():
    """rtk_layer_state mirrors _LayerStore: the library sees the buffers only when both are dense blocks of one capacity,
    and the numbers Python reads are the ones the library advanced."""
    import retake.longvideo_cache as lc

    L = 640
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    cache = lc.build_kvcache(cfg(1))
    for c in range(2):
        q0, k0, v0 = projections(c, L, torch.bfloat16)
        cache.kvcache_compression = True
        assert cache.update_pre_rope(q0, k0, v0, 0, chunk_ids(c, L), rot, SEC) is not None
        st = cache._layers[0]
        assert st.c.pending == L and st.pending_keep == L // 4 and st.c.k == st.k.data_ptr() and st.c.cap == st.k.shape[2]
        cache.after_forward()
        assert st.pending == 0 and st.length == (c + 1) * (L // 4) == st.pos_len and st.c.mask is None
    cache.key_cache[0] = cache.key_cache[0][:, :, ::2]      # an external writer hands over a strided view
    assert cache._layers[0].c.cap == 0                       # ... which the library must not touch


# ---------------------------------------------------------------------------------------------------
# MA-LLM-hard in one pass (rtk_mallm_hard_chain) against the reference's loop of single steps
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sync", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("T,N,C,tgt", [(24, 5, 64, 11), (96, 7, 1280, 40), (64, 3, 1152, 1), (33, 4, 256, 32),
                                       (160, 12, 3584, 80)])
def test_mallm_hard_chain_equals_the_loop_of_steps(T, N, C, tgt, dtype, sync):
    """memory_bank_compress_MALLM_hard_to == `while T > tgt: bank = memory_bank_compress_MALLM_hard(bank)` bit for bit:
    the same frames dropped in the same order (visual_compression.py:50-83 looped as qwen2_vl.py:406-408)."""
    import retake.visual_compression as vc

    x = torch.from_numpy(synth.frames_video(T + N + C, T, N, C)).to(dev()).to(dtype)
    bank = x
    while bank.shape[1] > tgt:
        bank = vc.memory_bank_compress_MALLM_hard(bank, sync=sync)
    got = vc.memory_bank_compress_MALLM_hard_to(x, tgt, sync=sync)
    assert got.shape == bank.shape == (1, tgt, N, C) and got.dtype == dtype
    assert torch.equal(got, bank)
    assert vc.memory_bank_compress_MALLM_hard_to(x, T, sync=sync) is x      # nothing to drop


def test_mallm_hard_chain_reference_fixtures():
    """The reference's own MA-LLM-hard loops (fixtures recorded from the imported reference): the one-pass form drops
    the same frames - the outputs are pure frame copies, so fp32 banks are bit-equal to the reference's."""
    import golden_util as gu
    import retake.visual_compression as vc

    seen = 0
    for name in gu.names("mallm_hard_") + gu.names("fp16mallm_hard_"):
        g = gu.load(name)
        dt = str(g["dtype"]) if "dtype" in g.files else "fp16"
        x = g["x"]
        if dt == "bf16":
            xt = torch.from_numpy(x.view(np.int16)).to(dev()).view(torch.bfloat16)
        elif name.startswith("fp16"):
            xt = torch.from_numpy(x).to(dev()).view(torch.float16)
        else:
            xt = torch.from_numpy(x).to(dev())
        out = vc.memory_bank_compress_MALLM_hard_to(xt, int(g["tgt"]), sync=bool(g["sync"]))
        loop = xt
        while loop.shape[1] > int(g["tgt"]):
            loop = vc.memory_bank_compress_MALLM_hard(loop, sync=bool(g["sync"]))
        assert torch.equal(out, loop), name
        if xt.dtype == torch.float32:
            np.testing.assert_array_equal(out.cpu().numpy(), g["out"])
        seen += 1
    assert seen >= 3


def test_compress_memory_bank_dispatches_the_chain():
    """qwen2_vl._compress_memory_bank('MA-LLM-hard') on the GPU = the reference's while loop (:402-410)."""
    import retake.qwen2_vl as rq
    import retake.visual_compression as vc

    x = torch.from_numpy(synth.frames_video(5, 48, 6, 128)).to(dev()).bfloat16()
    got, mask = rq._compress_memory_bank(x, 20, "MA-LLM-hard", False, True)
    loop = x
    while loop.shape[1] > 20:
        loop = vc.memory_bank_compress_MALLM_hard(loop, sync=False)
    assert mask is None and torch.equal(got, loop)


def test_masked_columns_never_leak_uninitialised_partials():
    """Pass 2 skips the columns of key-patch tokens (the reference overwrites their score with 1.0, :272-274) and leaves
    their slots of the column partials unwritten.  With the scratch poisoned with NaN beforehand the scores must still be
    finite everywhere, 1.0 at the masked tokens, and the kept set that of a cache that computes every column."""
    import retake.longvideo_cache as lc

    L, layers = 640, 2
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    caches = [lc.build_kvcache(cfg(layers)), lc.build_kvcache(cfg(layers, skip_masked_columns=False))]
    results = []
    for cache in caches:
        got = []
        for c in range(2):
            mask = torch.from_numpy(np.random.default_rng(7 + c).uniform(size=L) < 0.4).to(dev())
            cache.keypatches_mask_chunk = mask
            cache.kvcache_compression = True
            pos = chunk_ids(c, L)
            for l in range(layers):
                q0, k0, v = synth.qkv_chunk(300 + 10 * c + l, Hq, Hkv, L, D)
                cache.shift_temporal_ids_(pos, l)
                q = synth.rope_forward(torch.from_numpy(q0).to(dev()), pos, rot, SEC).bfloat16()
                k = synth.rope_forward(torch.from_numpy(k0).to(dev()), pos, rot, SEC).bfloat16()
                cache.update(k, torch.from_numpy(v).to(dev()).bfloat16(), l,
                             {"query_states": q, "position_ids": pos, "rotary_emb": rot, "mrope_section": SEC})
                if c == 0 and l == 0:
                    b = cache._batch
                    b.partials.fill_(float("nan"))      # whatever torch.empty handed out, now certainly not a number
                    b.score.fill_(float("nan"))
            cache.after_forward()
            b = cache._batch
            for l in range(layers):
                s = b.score[l]
                assert torch.isfinite(s).all() and (s[mask] == 1.0).all()
                got.append((s.clone(), b.keep_idx[l].clone()))
        results.append(got)
    for (sa, ia), (sb, ib) in zip(*results):
        assert torch.equal(ia, ib) and torch.equal(sa, sb)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L", [640, 2304])
def test_queries_scored_in_place_equal_the_packed_copy(dtype, L):
    """Prologue route, 16-bit tensors: the chunk-batched passes reading q0 from the projection layout (row pitch Hq * D,
    one pointer per layer) give bit for bit the scores, kept sets and caches of the route that packs a copy first."""
    import retake.longvideo_cache as lc

    layers = 3
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    zc, packed = (lc.build_kvcache(cfg(layers, prologue_operands="pre_rope")) for _ in range(2))
    for c in range(2):
        mask = torch.from_numpy(np.random.default_rng(c).uniform(size=L) < 0.3).to(dev())
        pz, pp = chunk_ids(c, L), chunk_ids(c, L)
        for cache in (zc, packed):
            cache.keypatches_mask_chunk = mask
            cache.kvcache_compression = True
        for l in range(layers):
            q0, k0, v0 = projections(500 + 10 * c + l, L, dtype)
            qz = q0.clone()
            assert zc.update_pre_rope(qz, k0, v0, l, pz, rot, SEC) is not None              # fresh q_rot, q0 scored in place
            assert zc._batch.q_keep[l] is qz
            qq = q0.clone()
            assert packed.update_pre_rope(qq, k0, v0, l, pp, rot, SEC, query_out=qq) is not None   # rotated in place: packed q~
            assert packed._batch.q_keep[l] is None
        for cache in (zc, packed):
            cache.after_forward()
        assert all(t is None for t in zc._batch.q_keep)
        for l in range(layers):
            assert torch.equal(zc._batch.score[l], packed._batch.score[l])
            assert torch.equal(zc._batch.keep_idx[l], packed._batch.keep_idx[l])
    for l in range(layers):
        assert torch.equal(zc.key_cache[l], packed.key_cache[l]) and torch.equal(zc.value_cache[l], packed.value_cache[l])
        assert torch.equal(zc.position_cache[l], packed.position_cache[l])


def test_in_place_and_packed_queries_mixed_within_a_chunk():
    """Layers of one chunk may hand their queries over differently (fresh q_rot: scored where they lie; rotated in place:
    a packed copy): the flush launches the passes per run of layers that agree - same bits as the all-packed route.
    (tools/fuzz_gpu.py found the flush refusing such a chunk.)"""
    import retake.longvideo_cache as lc

    layers, L, dtype = 4, 640, torch.bfloat16
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    mixed, packed = (lc.build_kvcache(cfg(layers, prologue_operands="pre_rope")) for _ in range(2))
    pattern = [True, False, False, True]     # in place / packed per layer
    for c in range(2):
        pz, pp = chunk_ids(c, L), chunk_ids(c, L)
        for cache in (mixed, packed):
            cache.keypatches_mask_chunk = None
            cache.kvcache_compression = True
        for l in range(layers):
            q0, k0, v0 = projections(800 + 10 * c + l, L, dtype)
            qz = q0.clone()
            assert mixed.update_pre_rope(qz, k0, v0, l, pz, rot, SEC, query_out=None if pattern[l] else qz) is not None
            qq = q0.clone()
            assert packed.update_pre_rope(qq, k0, v0, l, pp, rot, SEC, query_out=qq) is not None
        assert [t is not None for t in mixed._batch.q_keep] == pattern
        for cache in (mixed, packed):
            cache.after_forward()
        for l in range(layers):
            assert torch.equal(mixed._batch.score[l], packed._batch.score[l])
            assert torch.equal(mixed._batch.keep_idx[l], packed._batch.keep_idx[l])
    for l in range(layers):
        assert torch.equal(mixed.key_cache[l], packed.key_cache[l]) and torch.equal(mixed.value_cache[l], packed.value_cache[l])


def test_llava_attention_patch_takes_the_prologue():
    """The patched Qwen2 attention of LLaVA-Video (llava_onevision.py:59-141) on the stand-in module of tests/glue_stubs.py
    (head_dim 16: the C update route that runs the score passes per unit), two 640-token chunks, 2-D ids, fp32: the fused
    prologue and the op-by-op route agree on the attention output, the cache ids / values (exact) and keys (1e-5); the
    ids tensor the caller handed over is left alone (the reference shifts a clone)."""
    import glue_stubs as gs
    import retake.llava_onevision as lo
    import retake.longvideo_cache as lc

    L = 640
    att = gs.StubAttention(0, 64, 4, 2, None, A, seed=3).to_device(dev()).eval()
    llm = types.SimpleNamespace(hidden_size=64, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=2)

    def make(**extra):
        kw = {"compression_ratio": 0.25, "compression_method": "pivotkv", "pos_embed_reforge": True}
        kw.update(extra)
        return lc.build_kvcache(types.SimpleNamespace(text_config=llm, longvideo_kwargs={
            "kvcache_compression": True, "kvcache_compression_kwargs": kw}))

    caches = [make(), make(one_call_update=False)]
    outs = [[], []]
    g = torch.Generator(device=dev()).manual_seed(11)
    with torch.no_grad():
        for c in range(2):
            x = torch.randn((1, L, 64), generator=g, device=dev()) * 0.5
            ids = chunk_ids(c, L, mrope=False)
            keepsake = ids.clone()
            mask4 = gs.causal_mask(L, (c * (L // 4)) + L).to(dev())
            for i, cache in enumerate(caches):
                cache.kvcache_compression = True
                cache.keypatches_mask_chunk = None
                o = lo.retake_Qwen2Attention_forward(att, x, None, mask4, cache, None, position_ids=ids)
                cache.after_forward()
                outs[i].append(o[0].float())
                assert torch.equal(ids, keepsake)
    assert caches[0]._layers[0].length == 2 * (L // 4)
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())
    assert torch.equal(caches[0].position_cache[0], caches[1].position_cache[0])
    assert torch.equal(caches[0].value_cache[0], caches[1].value_cache[0])
    assert (caches[0].key_cache[0] - caches[1].key_cache[0]).abs().max().item() <= 1e-5


@pytest.mark.parametrize("llava,expanded", [(False, False), (True, False), (False, True)])
def test_whole_sequence_through_the_fused_prologues(llava, expanded):
    """text(5) -> video chunk(640) x 2 -> text(3) -> decode(1) x 2 through the patched attention of both models on the
    stand-in module, two layers, fp32: with the fused prologues (update_pre_rope for the chunks, append_pre_rope for text
    and decode) and with the op-by-op route the attention outputs agree, the ids tensor the caller handed over is in the
    same state after every layer call (shifted in place for Qwen2-VL, untouched for LLaVA), and the final caches hold the
    same ids / values (exact) and keys (1e-5).
    expanded: the text / decode ids of Qwen2-VL the way HF builds them - ONE row seen three times, `.unsqueeze(0).expand(3,
    -1, -1)` (qwen2_vl.py:589, strides (0, n, 1)): the reference's in-place shift of row 0 then moves t, h and w together,
    and so must the fused append (it once refused such ids with ValueError)."""
    import glue_stubs as gs
    import retake.llava_onevision as lo
    import retake.longvideo_cache as lc
    import retake.qwen2_vl as rq

    layers = [gs.StubAttention(l, 64, 4, 2, None if llava else (2, 3, 3), A, seed=9).to_device(dev()).eval() for l in range(2)]
    llm = types.SimpleNamespace(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2)

    def make(**extra):
        kw = {"compression_ratio": 0.25, "compression_method": "pivotkv", "pos_embed_reforge": True}
        kw.update(extra)
        lk = {"kvcache_compression": True, "kvcache_compression_kwargs": kw}
        if llava:
            return lc.build_kvcache(types.SimpleNamespace(text_config=llm, longvideo_kwargs=lk))
        cfg_ = types.SimpleNamespace(**vars(llm))
        cfg_.longvideo_kwargs = lk
        return lc.build_kvcache(cfg_)

    caches = [make(), make(one_call_update=False)]
    taken = []
    fused_append = caches[0].append_pre_rope

    def counting(*a, **k):
        out = fused_append(*a, **k)
        taken.append(out is not None)
        return out

    caches[0].append_pre_rope = counting
    steps = [("text", 5), ("video", 640), ("video", 640), ("text", 3), ("text", 1), ("text", 1)]
    g = torch.Generator(device=dev()).manual_seed(21)
    total = [0, 0]
    t_next = 0
    with torch.no_grad():
        for si, (kind, n) in enumerate(steps):
            x = torch.randn((1, n, 64), generator=g, device=dev()) * 0.5
            if llava:
                ids0 = (torch.arange(n, device=dev()) + 1000 * si).view(1, n)        # discontinuous on purpose
            elif kind == "video":
                ids0 = torch.from_numpy(synth.mrope_position_ids(50 * si, n // 64, 8, 8, hw0=2)).to(dev())
            else:
                ids0 = (torch.arange(n, device=dev()) + 1000 * si).view(1, 1, n).repeat(3, 1, 1)
            outs, ids_after = [], []
            for i, cache in enumerate(caches):
                cache.kvcache_compression = kind == "video"
                cache.keypatches_mask_chunk = None
                ids = ids0.clone()
                if expanded and kind == "text":
                    ids = ids0[0].clone().unsqueeze(0).expand(3, -1, -1)
                    assert ids.stride(0) == 0
                    prev_t = cache.get_prev_temporal_idx(0)
                    prev_t = int(prev_t)
                klen = cache.get_seq_length(0) + n
                mask4 = gs.causal_mask(n, klen).to(dev())
                per_layer = []
                for l, att in enumerate(layers):
                    if llava:
                        o = lo.retake_Qwen2Attention_forward(att, x, None, mask4, cache, None, position_ids=ids)
                    else:
                        o = rq.retake_Qwen2VLAttention_forward(att, x, mask4, ids, cache, False, True, None)
                    per_layer.append((o[0].float(), ids.clone()))
                    if expanded and kind == "text":   # the reference's semantics, restated: all three rows follow the cache
                        want = (torch.arange(n, device=dev()) + prev_t + 1).view(1, 1, n).expand(3, -1, -1)
                        assert torch.equal(ids, want), (si, l, i)
                        assert torch.equal(cache.position_cache[l][..., -n:], want)
                if kind == "video":
                    cache.after_forward()
                    # a chunk's fused prologue leaves the caller's ids to the flush (the last layer's shift, what the
                    # reference's layer loop leaves behind): compare the state after the chunk
                    per_layer = [(o_, ids.clone()) for o_, _ in per_layer]
                outs.append(per_layer)
            for (oa, ia), (ob, ib) in zip(*outs):
                assert (oa - ob).abs().max().item() <= 1e-4 * max(1.0, ob.abs().max().item()), (si, kind)
                assert torch.equal(ia, ib), (si, kind, "ids after the layer call / the chunk")
                if llava:
                    assert torch.equal(ia, ids0)
    for l in range(2):
        a, b = caches
        assert a.key_cache[l].shape == b.key_cache[l].shape == (1, 2, 5 + 160 + 160 + 3 + 2, 16)
        assert torch.equal(a.position_cache[l], b.position_cache[l])
        assert torch.equal(a.value_cache[l], b.value_cache[l])
        assert (a.key_cache[l] - b.key_cache[l]).abs().max().item() <= 1e-5
    assert caches[0].num_evicted_tokens == caches[1].num_evicted_tokens
    assert len(taken) == 4 * 2 and all(taken), "a text / decode segment fell off the fused append"


def test_native_rope_snapshot_follows_the_rotary_module():
    """The cache snapshots a rotary module's inv_freq once (native RoPE); a module whose inv_freq is then written in place
    or re-assigned must be snapshotted again - the prologue's rotated queries follow the module's CURRENT frequencies."""
    import retake.longvideo_cache as lc

    L, dtype = 640, torch.float32
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    cache = lc.build_kvcache(cfg(1))
    for c, change in enumerate((None, "in place", "re-assigned")):
        if change == "in place":
            rot.inv_freq.mul_(0.5)
        elif change == "re-assigned":
            rot.inv_freq = (rot.inv_freq * 3.0).contiguous()
        q0, k0, v0 = projections(70 + c, L, dtype)
        qk = q0.clone()
        pos = chunk_ids(c, L)
        prev = cache.get_prev_temporal_idx(0)
        cache.kvcache_compression = True
        out = cache.update_pre_rope(q0, k0, v0, 0, pos, rot, SEC, query_out=q0)
        assert out is not None
        want_ids = pos.clone()
        want_ids[0, 0] += int(prev) + 1 - want_ids[0, 0, 0]
        cos, sin = native_tables(want_ids, rot, dtype)
        assert torch.equal(out[0], (qk * cos) + (rot_half(qk) * sin)), change
        cache.after_forward()


# ---------------------------------------------------------------------------------------------------
# The route the product's own attention patch takes (update_pre_rope on the PRE-RoPE projections) against the REFERENCE:
#   * its fp32 goldens (the reference was handed the rotated tensors; here the library gets q0 / k0 of the same seed), and
#   * its own run on a bf16 model from the same bf16 q0 / k0 (fixtures pivotkv_prerope_bf16_*: the reference's rotation
#     helper on bf16 tensors, then PivotKVCache.update, longvideo_cache.py:35-116, :217-323).
# ---------------------------------------------------------------------------------------------------
import golden_util as gu  # noqa: E402

PK_PROLOGUE = [n for n in gu.names("pivotkv_") if not n.startswith(("pivotkv_bf16_", "pivotkv_fp16_", "pivotkv_prerope_"))
               and int(gu.load(n)["L"]) >= 512 and bool(gu.load(n)["reforge"]) and not bool(gu.load(n)["tie_case"])]


def _fixture_cache(g, n_layers, **extra):
    import retake.longvideo_cache as lc

    Hq_, Hkv_, D_ = (int(g[k]) for k in ("Hq", "Hkv", "D"))
    kw = {"compression_ratio": float(g["ratio"]), "compression_method": "pivotkv", "pos_embed_reforge": True}
    kw.update(extra)
    llm = types.SimpleNamespace(hidden_size=Hq_ * D_, num_hidden_layers=n_layers, num_attention_heads=Hq_,
                                num_key_value_heads=Hkv_)
    lk = {"kvcache_compression": True, "kvcache_compression_kwargs": kw}
    if len(g["mrope_section"]) == 0:      # LLaVA: the LLM's config sits under text_config (longvideo_cache.py:124)
        return lc.build_kvcache(types.SimpleNamespace(text_config=llm, longvideo_kwargs=lk))
    llm.longvideo_kwargs = lk
    return lc.build_kvcache(llm)


def _as_projection(x: torch.Tensor) -> torch.Tensor:
    """[1, H, L, D] values in the memory layout q_proj / k_proj / v_proj leave: [1, L, H*D], viewed [1, H, L, D]."""
    return x.transpose(1, 2).contiguous().transpose(1, 2)


@pytest.mark.parametrize("name", PK_PROLOGUE)
def test_reference_fp32_goldens_through_the_prologue(name):
    """Every fp32 golden whose chunks the prologue serves (L >= 512, reforging), all its chunks: update_pre_rope on the
    q0 / k0 the fixture's rotated inputs were made from -> kept indices bit-exact, kept V and ids exact, kept K within
    1e-5, scores within 5e-6 of the reference's - the same bars the reference-protocol route meets (test_hip_parity)."""
    g = gu.load(name)
    Hq_, Hkv_, D_, L, keep, layer = (int(g[k]) for k in ("Hq", "Hkv", "D", "L", "keep", "layer"))
    sec = [int(s) for s in g["mrope_section"]] or None
    cache = _fixture_cache(g, layer + 1)
    rot = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]), device=dev())
    for c in range(int(g["n_chunks"])):
        pre = f"c{c}_"
        q0, k0, v = synth.qkv_chunk(int(g["seed"]) * 100 + c, Hq_, Hkv_, L, D_)
        mask = g[pre + "mask"]
        cache.keypatches_mask_chunk = torch.from_numpy(mask).to(dev()) if mask.size else None
        cache.kvcache_compression = True
        # the fixture stores the ids after the attention patch's continuity shift; the prologue's own shift of them is 0
        ids = torch.from_numpy(g[pre + "pos"]).to(dev())
        out = cache.update_pre_rope(_as_projection(torch.from_numpy(q0).to(dev())), _as_projection(torch.from_numpy(k0).to(dev())),
                                    _as_projection(torch.from_numpy(v).to(dev())), layer, ids, rot, sec,
                                    shift_ids_in_place=sec is not None)
        assert out is not None, "the prologue declined a golden chunk"
        assert torch.equal(cache._batch.pos_old[layer].reshape(ids.shape), torch.from_numpy(g[pre + "pos"]).to(dev()))
        score = None if cache._batch.keep_all else cache.last_scores.cpu().numpy()
        idx = cache.last_keep_indices.cpu().numpy()
        cache.after_forward()
        if score is not None:
            assert np.abs(score - g[pre + "score32"]).max() < 5e-6 and np.abs(score - g[pre + "score64"]).max() < 5e-6
        np.testing.assert_array_equal(idx, g[pre + "keep_idx"])
        kk = cache.key_cache[layer][:, :, -keep:].cpu().numpy()
        err = np.abs(kk - g[pre + "kept_k"]).max()
        assert err <= 1e-5, err
        assert synth.checksum(cache.value_cache[layer][:, :, -keep:].cpu().numpy()) == int(g[pre + "kept_v_crc"])
        np.testing.assert_array_equal(cache.position_cache[layer].cpu().numpy(), g[pre + "position_cache"])
        assert cache.num_evicted_tokens[layer] == int(g[pre + "num_evicted"])
        print(f"\n[{name} c{c}] prologue route: indices exact, max |K - reference| {err:.2e}")


def _bf(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16)


@pytest.mark.parametrize("mode", ["reference/fp32", "reference/reference", "pre_rope/fp32"])
@pytest.mark.parametrize("name", gu.names("pivotkv_prerope_bf16_"))
def test_prologue_bf16_against_the_reference_run_from_pre_rope_projections(name, mode):
    """Production dtype, the route the product's attention patch takes.  The reference was run on a bf16 model's tensors:
    bf16 q0 / k0 -> its rotation helper in bf16 -> PivotKVCache.update (un-rotation, bf16 logits / probabilities / sums,
    topk, re-rotation).  update_pre_rope gets the same q0 / k0 / v and unshifted ids, three layers per chunk, every chunk:
      ids      the shifted ids equal the reference's; the new ids of the kept tokens equal its position cache;
      q_rot    equals the reference helper's bf16 rotation except where a table entry's fp32 value straddles a bf16
               midpoint (correctly rounded vs libm sin / cos); counted;
      kept V   exact copies of the rows the product's indices name.
    mode = prologue_operands / score_rounding:
      reference/fp32 (the DEFAULT)  operands = the reference's round-tripped q~ / k~; scores fp32-accurate on them (within
               2e-5 of their exact score); every token the kept sets disagree on has a reference score within ONE bf16 ulp of
               the reference's threshold (the quantisation its own scores carry); kept K bit-exact where ids agree;
      reference/reference           + the reference's bf16 rounding chain: scores equal its bf16 scores up to isolated 1-ulp
               entries, kept set equal up to exact ties (test_oracle_golden.check_bf16_against_reference);
      pre_rope/fp32 (opt-in)        operands = q0 / k0: scores within 2e-5 of THEIR exact score, which differs from the
               round-tripped operands' by up to E ~ 2e-2 (6 bf16 ulps): a token may change sides only if its reference
               score lies within 2E of the reference's threshold (order statistics are 1-Lipschitz); counted; kept K is one
               rotation of k0 - compared in bf16 ulps and against the exact rotation (closer than the reference's)."""
    import test_oracle_golden as tog

    operands, rounding = mode.split("/")
    g = gu.load(name)
    if int(g["L"]) < 512 and operands == "pre_rope":
        pytest.skip("chunks under the prologue's gate take the op-by-op route: the reference's operands whatever the option")
    Hq_, Hkv_, D_, L, keep = (int(g[k]) for k in ("Hq", "Hkv", "D", "L", "keep"))
    sec = [int(s) for s in g["mrope_section"]] or None
    a_scale = float(g["attention_scaling"])
    n_layers = 3
    cache = _fixture_cache(g, n_layers, prologue_operands=operands, score_rounding=rounding)
    rot = synth.RotaryStub(g["inv_freq"], a_scale, device=dev())
    # measured on MI355X (profiles/r14_parity_stats.txt); the bars are twice the measurement (at least 2)
    XOR_BAR = {"reference": {256: 2, 1568: 12, 2304: 28, 6272: 48}, "pre_rope": {256: 4, 1568: 24, 2304: 24, 6272: 80}}[operands][L]
    for c in range(int(g["n_chunks"])):
        pre = f"c{c}_"
        q0b, k0b, vb, pos_in, pos, mask = gu.pivotkv_prerope_chunk_inputs(g, c)
        q0, k0, v = (_as_projection(_bf(x).to(dev())) for x in (q0b, k0b, vb))
        cache.keypatches_mask_chunk = torch.from_numpy(mask).to(dev())
        cache.kvcache_compression = True
        ids = torch.from_numpy(pos_in).to(dev())
        q_rots = []
        for l in range(n_layers):
            ql = q0.clone()     # (the rotated queries may be written over the projection, as in the model)
            out = cache.update_pre_rope(ql, k0, v, l, ids, rot, sec, shift_ids_in_place=sec is not None)
            if L < 512:
                # chunks below the prologue's gate: the patch's op-by-op route (qwen2_vl.py:68-86 as the build restates it in
                # retake/qwen2_vl.py:_qkv_and_cache_update) - id shift on the device, rotary module, the rotation helper, update
                assert out is None
                import retake.longvideo_cache as lc

                cache.shift_temporal_ids_(ids, l)
                cos, sin = rot(v, ids)
                qr, kr = lc.apply_multimodal_rotary_pos_emb(ql, k0, cos, sin, sec) if sec else lc.apply_rotary_pos_emb(ql, k0, cos, sin)
                cache.update(kr, v, l, {"query_states": qr, "position_ids": ids, "rotary_emb": rot, "mrope_section": sec})
                q_rots.append(qr)
                continue
            assert out is not None, "the prologue declined a fixture chunk"
            q_rots.append(out[0])
        b = cache._batch
        assert b.batched_passes == (L >= 512) and len(b.pending) == n_layers
        want_ids = torch.from_numpy(pos).to(dev())
        for l in range(n_layers):
            shifted = b.pos_old[l].reshape(want_ids.shape) if L >= 512 else ids   # (small chunks: shifted in place, eagerly)
            assert torch.equal(shifted, want_ids), "continuity shift differs from the reference's"
        cache.after_forward()
        # --- rotated queries against the reference helper's (restated on the CPU, crc-pinned to the generator's tensors)
        qr_ref, _ = gu.rotate_like_a_bf16_model(g, c, q0b, k0b)
        qr = q_rots[0].cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
        n_q = int((qr != qr_ref).sum())
        dq = np.abs(orc.bf16_bits_to_f32(qr) - orc.bf16_bits_to_f32(qr_ref))
        assert (dq <= gu.bf16_ulp(orc.bf16_bits_to_f32(qr_ref))).all() and n_q <= max(8, qr.size // 20000)
        ref = orc.bf16_bits_to_f32(g[pre + "score_bf16"])
        ref_idx = g[pre + "keep_idx"]
        thr = np.sort(ref)[::-1][keep - 1]
        ref_pos = g[pre + "position_cache"][..., -keep:].reshape(-1, keep)
        s64 = g[pre + ("score64" if operands == "reference" else "score64_pre")].copy()
        s64[mask] = 1.0
        E = np.abs(ref - s64).max()      # how far the reference's bf16 scores are from the exact score of these operands
        for l in range(n_layers):
            score = b.score[l].cpu().numpy()
            idx = b.keep_idx[l].cpu().numpy()
            vv = cache.value_cache[l][:, :, -keep:].cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
            assert np.array_equal(vv[0], vb[0][:, idx])
            pos_new = cache.position_cache[l][..., -keep:].cpu().numpy().reshape(-1, keep)
            kk = cache.key_cache[l][:, :, -keep:].cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
            xor = np.setxor1d(idx, ref_idx)
            # the new ids are the reference's formula on the product's own kept set (longvideo_cache.py:283-295): gather, then
            # on the temporal row t' = tmin + trunc((t - tmin) * float32(keep / L)), fp32 multiply
            pk = pos.reshape(-1, L)[:, idx].copy()
            pk[0] = pk[0].min() + ((pk[0] - pk[0].min()).astype(np.float32) * np.float32(keep / L)).astype(np.int64)
            np.testing.assert_array_equal(pos_new, pk)
            # (plain-RoPE ids are one per token: tmin is the id of the FIRST kept token, so a kept set that differs there
            # moves every new id - then only the sets are comparable with the reference's, not ids / keys)
            same_tmin = pos.reshape(-1, L)[0, idx].min() == pos.reshape(-1, L)[0, ref_idx].min()
            if l:   # identical inputs: identical layers, bit for bit
                assert torch.equal(b.score[l], b.score[0]) and torch.equal(cache.key_cache[l], cache.key_cache[0])
            if rounding == "reference" and not same_tmin:
                assert (score != ref).sum() <= 4 and (np.abs(ref[xor] - thr) <= gu.bf16_ulp(np.full(xor.size, thr))).all()
                print(f"\n[{name} c{c}] {mode}: kept xor {xor.size} (ties) includes the first kept token: new ids not comparable")
                continue
            if rounding == "reference":
                nbad, nxor, a, bb = tog.check_bf16_against_reference(g, c, score, idx, kk, pos_new, "prologue, reference rounding",
                                                                     max_bad={256: 1, 1568: 2, 2304: 4, 6272: 10}.get(L))   # measured 0 / 0 / 0-1 / 1-5: twice that
                np.testing.assert_array_equal(a, bb)
                if l == 0:
                    print(f"\n[{name} c{c}] {mode}: {nbad} of {L} scores differ from the reference's by one bf16 ulp, kept xor {nxor} "
                          f"(ties), kept K bit-exact; q_rot: {n_q} of {qr.size} entries differ by one bf16 ulp")
                continue
            err = np.abs(score - s64).max()
            assert err < 2e-5, err
            if operands == "reference":
                assert (np.abs(ref[xor] - thr) <= gu.bf16_ulp(np.full(xor.size, thr))).all(), \
                    "kept sets differ beyond the reference's own score quantisation"
            else:
                assert (np.abs(ref[xor] - thr) <= 2 * E + 4e-5).all(), "a token changed sides farther from the threshold than 2E"
            assert xor.size <= XOR_BAR, (xor.size, XOR_BAR)
            common, ia, ib = np.intersect1d(idx, ref_idx, return_indices=True)
            same_pos = (pos_new[:, ia] == ref_pos[:, ib]).all(0)
            if not same_tmin:
                print(f"\n[{name} c{c}] {mode}: {xor.size // 2} of {keep} kept tokens differ (xor {xor.size}), the first kept token among "
                      f"them: new ids not comparable with the reference's")
                continue
            assert same_pos.mean() > 0.9
            if xor.size == 0:
                np.testing.assert_array_equal(pos_new, ref_pos)
            mine_bits = kk[0][:, ia[same_pos]]
            theirs_bits = g[pre + "kept_k_bits"][0][:, ib[same_pos]]
            if operands == "reference":
                np.testing.assert_array_equal(mine_bits, theirs_bits)
                if l == 0:
                    print(f"\n[{name} c{c}] {mode} (product default): {xor.size // 2} of {keep} kept tokens differ (xor {xor.size}), all "
                          f"within one bf16 ulp of the reference's threshold {thr}; max |score - exact| {err:.2e}; kept K of "
                          f"{int(same_pos.sum())} common tokens bit-exact; q_rot: {n_q} of {qr.size} entries differ by one bf16 ulp")
                continue
            mine = orc.bf16_bits_to_f32(mine_bits).astype(np.float64)
            theirs = orc.bf16_bits_to_f32(theirs_bits).astype(np.float64)
            # exact: a * (k0 * cos + rotate_half(k0) * sin) at the new ids, fp64 tables
            k0f = orc.bf16_bits_to_f32(k0b)[0].astype(np.float64)                     # [Hkv, L, D]
            inv = g["inv_freq"].astype(np.float64)
            tok = common[same_pos]
            pn = ref_pos[:, ib[same_pos]].astype(np.float64)                       # [P, n]
            if sec:
                rows = np.concatenate([np.full(sz, i % 3) for i, sz in enumerate(sec * 2)])[: D_ // 2]
                ang = pn[rows].T * inv[None, :]                                    # [n, D/2]: the id row of the channel's section
            else:
                ang = pn[0][:, None] * inv[None, :]
            cosx, sinx = np.cos(ang) * a_scale, np.sin(ang) * a_scale
            x = k0f[:, tok]                                                       # [Hkv, n, D]
            x1, x2 = x[..., : D_ // 2], x[..., D_ // 2:]
            exact = np.concatenate([x1 * cosx - x2 * sinx, x2 * cosx + x1 * sinx], -1)
            e_mine, e_theirs = np.abs(mine - exact).mean(), np.abs(theirs - exact).mean()
            # a rotation mixes a channel with its partner: the roundings scale with the pair's norm, not the element's value
            pair = np.sqrt(x1 ** 2 + x2 ** 2) * a_scale
            ulp = gu.bf16_ulp(np.maximum(np.concatenate([pair, pair], -1), 2.0 ** -6).astype(np.float32))
            worst = (np.abs(mine - theirs) / ulp).max()
            frac_equal = (mine == theirs).mean()
            assert e_mine <= e_theirs, (e_mine, e_theirs)
            assert worst <= 8.0, worst
            if l == 0:
                far = int((np.abs(ref[xor] - thr) > gu.bf16_ulp(np.full(xor.size, thr))).sum())
                print(f"\n[{name} c{c}] {mode} (opt-in): {xor.size // 2} of {keep} kept tokens differ (xor {xor.size}; {far} of them "
                      f"farther than one bf16 ulp from the reference's threshold {thr}, all within 2E = {2 * E:.3f}); max |score - "
                      f"exact pre-RoPE| {err:.2e}; kept K of {int(same_pos.sum())} common tokens: {100 * frac_equal:.1f} % bit-equal, "
                      f"worst {worst:.1f} bf16 ulp (of the pair norm) apart, mean |K - exact| {e_mine:.3e} (prologue) vs {e_theirs:.3e} (reference)")
        if not torch.equal(cache.position_cache[0][..., -1:].cpu(), torch.from_numpy(g[pre + "position_cache"][..., -1:])):
            break   # the next chunk's continuity shift starts from another id than the reference's did


@pytest.mark.parametrize("rounding", ["fp32", "reference", "fast"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("mrope", [True, False])
def test_prologue_with_reference_operands_equals_the_update_route_bitwise(dtype, rounding, mrope):
    """The product-default route (update_pre_rope, prologue_operands="reference") and the reference's protocol (`update` on
    tensors rotated op by op with the same tables) are the SAME computation: identical scoring operands, hence identical
    scores, kept sets, kept keys, values and ids - bit for bit, in every dtype and score arithmetic, M-RoPE and plain ids,
    two chunks x three layers, key-patch mask on."""
    import retake.longvideo_cache as lc

    if rounding != "fp32" and dtype == torch.float32:
        pytest.skip("score_rounding applies to 16-bit tensors")
    if rounding == "fast" and dtype == torch.float16:
        pytest.skip("the fast mode is a bf16 mode")
    layers, L = 3, 640
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    sec = SEC if mrope else None
    pro = lc.build_kvcache(cfg(layers, score_rounding=rounding))
    upd = lc.build_kvcache(cfg(layers, score_rounding=rounding))
    for c in range(2):
        mask = torch.from_numpy(np.random.default_rng(40 + c).uniform(size=L) < 0.3).to(dev())
        ids_p, ids_u = chunk_ids(c, L, mrope), chunk_ids(c, L, mrope)
        for cache in (pro, upd):
            cache.keypatches_mask_chunk = mask
            cache.kvcache_compression = True
        for l in range(layers):
            q0, k0, v0 = projections(300 + 10 * c + l, L, dtype)
            out = pro.update_pre_rope(q0.clone(), k0, v0, l, ids_p, rot, sec, shift_ids_in_place=mrope)
            assert out is not None
            ids_l = ids_u if mrope else ids_u.clone()        # (LLaVA's patch shifts a clone per layer)
            upd.shift_temporal_ids_(ids_l, l)
            cos, sin = native_tables(ids_l, rot, dtype, mrope)
            qr = (q0 * cos) + (rot_half(q0) * sin)
            kr = (k0 * cos) + (rot_half(k0) * sin)
            assert torch.equal(out[0], qr)
            upd.update(kr, v0, l, {"query_states": qr, "position_ids": ids_l, "rotary_emb": rot, "mrope_section": sec})
        pb, ub = pro._batch, upd._batch
        pro.after_forward()
        upd.after_forward()
        for l in range(layers):
            assert torch.equal(pb.score[l], ub.score[l]), (c, l, "scores")
            assert torch.equal(pb.keep_idx[l], ub.keep_idx[l]), (c, l, "kept set")
    for l in range(layers):
        assert torch.equal(pro.key_cache[l], upd.key_cache[l]) and torch.equal(pro.value_cache[l], upd.value_cache[l])
        assert torch.equal(pro.position_cache[l], upd.position_cache[l])
    assert pro.num_evicted_tokens == upd.num_evicted_tokens


@pytest.mark.parametrize("mode", ["reference/fp32", "reference/reference", "pre_rope/fp32"])
@pytest.mark.parametrize("name", gu.names("pivotkv_prerope_fp16_"))
def test_prologue_fp16_against_the_reference_run_from_pre_rope_projections(name, mode):
    """The float16 twin: the reference run on an fp16 model's tensors from the fp16 pre-RoPE projections (its rotation helper
    in fp16, then PivotKVCache.update), against update_pre_rope on the same q0 / k0 / v, three layers per chunk.  Same
    statements as the bf16 test with fp16 ulps: ids equal; rotated queries equal to the helper's up to table midpoints;
    reference operands -> kept K bit-exact, kept sets differ only within one fp16 ulp of the reference's threshold (fp32
    score arithmetic) or by threshold ties (the reference's fp16 chain, RTK_F16_REFROUND); pre-RoPE operands -> within 2E."""
    operands, rounding = mode.split("/")
    g = gu.load(name)
    Hq_, Hkv_, D_, L, keep = (int(g[k]) for k in ("Hq", "Hkv", "D", "L", "keep"))
    sec = [int(s) for s in g["mrope_section"]] or None
    n_layers = 2
    cache = _fixture_cache(g, n_layers, prologue_operands=operands, score_rounding=rounding)
    rot = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]), device=dev())

    def f32(bits):
        return np.ascontiguousarray(bits).view(np.float16).astype(np.float32)

    def f16(bits):
        return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.float16)

    for c in range(int(g["n_chunks"])):
        pre = f"c{c}_"
        q0b, k0b, vb, pos_in, pos, mask = gu.pivotkv_prerope_chunk_inputs(g, c)
        q0, k0, v = (_as_projection(f16(x).to(dev())) for x in (q0b, k0b, vb))
        cache.keypatches_mask_chunk = torch.from_numpy(mask).to(dev())
        cache.kvcache_compression = True
        ids = torch.from_numpy(pos_in).to(dev())
        outs = [cache.update_pre_rope(q0.clone(), k0, v, l, ids, rot, sec) for l in range(n_layers)]
        assert all(o is not None for o in outs)
        b = cache._batch
        want_ids = torch.from_numpy(pos).to(dev())
        assert all(torch.equal(b.pos_old[l].reshape(want_ids.shape), want_ids) for l in range(n_layers))
        cache.after_forward()
        qr_ref, _ = gu.rotate_like_a_bf16_model(g, c, q0b, k0b)
        qr = outs[0][0].cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
        n_q = int((qr != qr_ref).sum())
        assert (np.abs(f32(qr) - f32(qr_ref)) <= gu.fp16_ulp(f32(qr_ref))).all() and n_q <= max(8, qr.size // 20000)
        ref = f32(g[pre + "score_bf16"])
        ref_idx = g[pre + "keep_idx"]
        thr = np.sort(ref)[::-1][keep - 1]
        ref_pos = g[pre + "position_cache"][..., -keep:].reshape(-1, keep)
        s64 = g[pre + ("score64" if operands == "reference" else "score64_pre")].copy()
        s64[mask] = 1.0
        E = np.abs(ref - s64).max()
        for l in range(n_layers):
            score, idx = b.score[l].cpu().numpy(), b.keep_idx[l].cpu().numpy()
            xor = np.setxor1d(idx, ref_idx)
            vv = cache.value_cache[l][:, :, -keep:].cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
            assert np.array_equal(vv[0], vb[0][:, idx])
            pos_new = cache.position_cache[l][..., -keep:].cpu().numpy().reshape(-1, keep)
            kk = cache.key_cache[l][:, :, -keep:].cpu().contiguous().view(torch.int16).numpy().view(np.uint16)[0]
            common, ia, ib = np.intersect1d(idx, ref_idx, return_indices=True)
            same_pos = (pos_new[:, ia] == ref_pos[:, ib]).all(0)
            assert same_pos.mean() > 0.9
            if rounding == "reference":
                bad = np.nonzero(score != ref)[0]
                assert np.array_equal(score.astype(np.float16).astype(np.float32), score) and bad.size <= 16, bad.size
                assert (np.abs(score[bad] - ref[bad]) <= gu.fp16_ulp(np.minimum(np.abs(score[bad]), np.abs(ref[bad])))).all()
                assert (np.abs(ref[xor] - thr) <= gu.fp16_ulp(np.full(xor.size, thr))).all()
                np.testing.assert_array_equal(kk[:, ia[same_pos]], g[pre + "kept_k_bits"][0][:, ib[same_pos]])
                note = f"{bad.size} of {L} scores off by one fp16 ulp, kept xor {xor.size}, kept K bit-exact"
            else:
                err = np.abs(score - s64).max()
                assert err < 2e-5, err
                if operands == "reference":
                    assert (np.abs(ref[xor] - thr) <= gu.fp16_ulp(np.full(xor.size, thr))).all()
                    np.testing.assert_array_equal(kk[:, ia[same_pos]], g[pre + "kept_k_bits"][0][:, ib[same_pos]])
                    note = f"{xor.size // 2} of {keep} kept tokens differ (within one fp16 ulp of the threshold), kept K bit-exact"
                else:
                    assert (np.abs(ref[xor] - thr) <= 2 * E + 4e-5).all()
                    note = f"{xor.size // 2} of {keep} kept tokens differ (all within 2E = {2 * E:.4f})"
                note += f"; max |score - exact| {err:.2e}"
            assert xor.size <= 16
            if l == 0:
                print(f"\n[{name} c{c}] {mode}: {note}; q_rot: {n_q} of {qr.size} entries differ by one fp16 ulp")
        if not torch.equal(cache.position_cache[0][..., -1:].cpu(), torch.from_numpy(g[pre + "position_cache"][..., -1:])):
            break


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("L,mrope", [(640, True), (2304, True), (6272, False)])
def test_the_next_layers_shift_rides_in_the_update_launch(L, mrope, dtype):
    """qwen2_vl.py:68-73 on the reference's protocol: every layer shifts the shared ids tensor to follow ITS last cached
    temporal id before its RoPE.  rtk_pivotkv_update(RTK_UPDATE_SHIFT_NEXT) applies layer l + 1's shift inside layer l's
    launch (after every workgroup has read the ids layer l works with), so `shift_temporal_ids_` launches once per chunk.
    The ids every layer sees, the ids left in the caller's tensor and the caches equal the launch-per-layer route's bit
    for bit; the number of shift launches is counted."""
    import retake._native as nv
    import retake.longvideo_cache as lc

    layers, n_chunks = 4, 3
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    sec = SEC if mrope else None
    g = torch.Generator(device=dev()).manual_seed(5)
    pool = [tuple((1.7 * torch.randn((1, h, L, D), generator=g, device=dev())).to(dtype) for h in (Hq, Hkv, Hkv)) for _ in range(3)]
    kid = nv.profile_kernel_ids()["position_shift"]

    def run(cache):
        seen = []
        nv.check(nv.lib.rtk_profile_reset(), "profile_reset")
        nv.check(nv.lib.rtk_profile_enable_mask(1 << kid), "profile_enable")
        for c in range(n_chunks):
            # (many temporal steps, so that the layers' last kept ids - the bases of their shifts - differ)
            pos = torch.from_numpy(synth.mrope_position_ids(10 + 7 * c, L // 4, 2, 2, hw0=2)).to(dev()) if mrope \
                else chunk_ids(c, L, False)
            cache.keypatches_mask_chunk = torch.from_numpy(np.random.default_rng(c).uniform(size=L) < 0.3).to(dev())
            cache.kvcache_compression = True
            for l in range(layers):
                q, k, v = pool[(c + l) % 3]
                cache.shift_temporal_ids_(pos, l)
                seen.append(pos.clone())
                # "shift_next_position_ids": the Qwen2-VL patch's opt-in (it shares `pos` between the layers)
                kw = {"query_states": q, "position_ids": pos, "rotary_emb": rot, "shift_next_position_ids": ask}
                if mrope:
                    kw["mrope_section"] = sec
                cache.update(k, v, l, kw)
                assert "shift_next_position_ids" not in kw and "position_ids" not in kw   # popped like the reference's keys
            seen.append(pos.clone())    # what the reference's loop leaves in the caller's tensor: the LAST layer's shift
            cache.after_forward()
        cache.check()                    # synchronises; raises if a bounded wait ran out
        nv.check(nv.lib.rtk_profile_enable(0), "profile_enable")
        n = nv.profile_read().get("position_shift", (0, 0.0))[0]
        tk = cache._batch.shift_ticket.cpu()
        assert int(tk[31]) == 0 and int(tk[32:].abs().max()) == 0   # no wait ran out; the arrival counters are back at zero
        assert int(cache._batch.shift_latch[0]) == 0                # ... nor was the host-visible word touched
        return seen, n, int(tk[0])   # launches that carried a shift

    ask = True
    fused, apart = lc.build_kvcache(cfg(layers)), lc.build_kvcache(cfg(layers, shift_next_in_update=False))
    seen_f, n_f, epochs_f = run(fused)
    seen_a, n_a, epochs_a = run(apart)
    assert n_a == n_chunks * layers and epochs_a == 0
    # every layer's shift was either its own launch or rode in the previous layer's update (layer 0's never does; the
    # first chunk creates the layers' stores on the general route, a growing store re-binds: a launch per layer there).
    # fp32 chunks run their score passes inside update, where the library could still decline AFTER the launch: they never
    # carry the shift
    assert n_f + epochs_f == n_chunks * layers, (n_f, epochs_f)
    assert epochs_f >= layers - 1 if dtype is not torch.float32 else epochs_f == 0, (n_f, epochs_f)
    # a caller that does NOT ask (the reference's own cache_kwargs; the LLaVA patch) never has its ids written by update
    ask = False
    plain = lc.build_kvcache(cfg(layers))
    seen_p, n_p, epochs_p = run(plain)
    assert n_p == n_chunks * layers and epochs_p == 0
    assert all(torch.equal(a, b) for a, b in zip(seen_p, seen_a))
    assert len(seen_f) == len(seen_a)
    for i, (a, b) in enumerate(zip(seen_f, seen_a)):
        assert torch.equal(a, b), f"ids differ at step {i}"
    for l in range(layers):
        assert torch.equal(fused.key_cache[l], apart.key_cache[l])
        assert torch.equal(fused.value_cache[l], apart.value_cache[l])
        assert torch.equal(fused.position_cache[l], apart.position_cache[l])
    # the layers' last ids differ (each keeps its own tokens), so the shifts were real ones
    assert len({int(fused.position_cache[l][0, 0, -1] if mrope else fused.position_cache[l][0, -1]) for l in range(layers)}) > 1


def test_a_preshifted_tensor_is_recognised_only_as_itself():
    """The memo of the in-launch shift names the tensor OBJECT, its version, the layer and the stream: a clone, a tensor
    modified in between, another layer or a skipped layer all get their own launch (which is idempotent)."""
    import retake._native as nv
    import retake.longvideo_cache as lc

    layers, L = 4, 640
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    g = torch.Generator(device=dev()).manual_seed(6)
    q, k, v = ((1.7 * torch.randn((1, h, L, D), generator=g, device=dev())).to(torch.bfloat16) for h in (Hq, Hkv, Hkv))
    cache = lc.PivotKVCache(cfg(layers), reserve_tokens=8 * L)   # (no store grows inside the test: growth re-binds on the general route)
    kid = nv.profile_kernel_ids()["position_shift"]

    def chunk(c, between):
        pos = chunk_ids(c, L)
        cache.kvcache_compression = True
        cache.keypatches_mask_chunk = None
        launches = []
        for l in range(layers):
            nv.check(nv.lib.rtk_profile_reset(), "profile_reset")
            nv.check(nv.lib.rtk_profile_enable_mask(1 << kid), "profile_enable")
            pos = between(l, pos)
            cache.shift_temporal_ids_(pos, l)
            torch.cuda.synchronize()
            nv.check(nv.lib.rtk_profile_enable(0), "profile_enable")
            launches.append(nv.profile_read().get("position_shift", (0, 0.0))[0])
            prev = cache.get_prev_temporal_idx(l)
            assert int(pos[0, 0, 0]) == int(prev) + 1
            cache.update(k, v, l, {"query_states": q, "position_ids": pos, "rotary_emb": rot, "mrope_section": SEC,
                                   "shift_next_position_ids": True})
        cache.after_forward()
        return launches

    chunk(0, lambda l, p: p)                                          # creates the layers' stores (general route)
    assert chunk(1, lambda l, p: p) == [1, 0, 0, 0]
    assert chunk(2, lambda l, p: p.clone()) == [1, 1, 1, 1]           # LLaVA's patch: a clone per layer
    assert chunk(3, lambda l, p: p.add_(0) if l == 2 else p) == [1, 0, 1, 0]   # touched through torch: version moved


def test_in_launch_shift_that_runs_out_shifts_nothing_and_raises():
    """The watcher's wait is bounded by polls; when it runs out (forced: one arrival counter pre-loaded so the total can never
    match - what sharing the words between two streams would do) the launch must NOT rewrite the ids and must NOT zero the
    counters: it latches ticket[31] and the host-visible word, the watcher of the next launch returns at once, and the
    cache raises at its next entry point (and from check()).  Afterwards the cache shifts with one launch per layer again
    and a fresh cache is unaffected."""
    import time

    import retake.longvideo_cache as lc

    layers, L = 4, 640
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    g = torch.Generator(device=dev()).manual_seed(7)
    q, k, v = ((1.7 * torch.randn((1, h, L, D), generator=g, device=dev())).to(torch.bfloat16) for h in (Hq, Hkv, Hkv))
    cache = lc.PivotKVCache(cfg(layers), reserve_tokens=8 * L)

    def kw(pos):
        return {"query_states": q, "position_ids": pos, "rotary_emb": rot, "mrope_section": SEC, "shift_next_position_ids": True}

    def chunk(c):
        pos = chunk_ids(c, L)
        cache.kvcache_compression, cache.keypatches_mask_chunk = True, None
        for l in range(layers):
            cache.shift_temporal_ids_(pos, l)
            cache.update(k, v, l, kw(pos))
        cache.after_forward()

    chunk(0)            # creates the stores (general route)
    chunk(1)            # steady state: the shifts ride in the update launches
    cache.check()
    b = cache._batch
    assert int(b.shift_ticket[0]) >= layers - 1 and int(b.shift_latch[0]) == 0
    # --- force the run-out: counter 0 can never equal its share of the workgroups
    launches_before = int(b.shift_ticket[0])
    b.shift_ticket[32] = 1000
    pos = chunk_ids(2, L)
    cache.kvcache_compression = True
    cache.shift_temporal_ids_(pos, 0)
    before = pos.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cache.update(k, v, 0, kw(pos))                       # its watcher polls, gives up, latches - seconds later
    # the host is ahead of the device: it still believes layer 1's shift rode in that launch (no launch of its own) and
    # issues layer 1's update on the same words - whose watcher must return AT ONCE on the latched words, shifting nothing
    cache.shift_temporal_ids_(pos, 1)
    cache.update(k, v, 1, kw(pos))
    torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    assert torch.equal(pos, before), "a failed wait must leave the ids untouched"
    tk = b.shift_ticket.cpu()
    assert int(tk[31]) == 1 and int(b.shift_latch[0]) == 1, (int(tk[31]), int(b.shift_latch[0]))   # device latch + pinned host word, once
    assert int(tk[32]) > 1000 and int(tk[33:].abs().max()) > 0, "a failed wait must leave the counters alone"
    assert int(tk[0]) == launches_before                 # neither launch counted itself as having shifted
    assert 0.05 < waited < 30, waited                    # bounded: seconds, not a hung queue
    # --- the host raises at its next entry point, without having been asked to synchronise ...
    with pytest.raises(RuntimeError, match="ran out of its bounded wait"):
        cache.shift_temporal_ids_(pos, 2)
    # ... has put the words back, and has switched the feature off for this cache
    assert int(b.shift_latch[0]) == 0 and int(b.shift_ticket.abs().max()) == 0 and cache.shift_next_in_update is False
    cache.check()
    # a fresh cache on the same device is unaffected
    other = lc.PivotKVCache(cfg(layers), reserve_tokens=8 * L)
    cache = other
    chunk(0)
    chunk(1)
    other.check()
    assert int(other._batch.shift_ticket[0]) >= layers - 1


def test_a_call_the_one_launch_route_declines_falls_back_also_when_it_asked_for_the_shift():
    """rtk_pivotkv_update declines (RTK_EUNSUPPORTED) before launching anything - here: a query tensor whose address is not
    16-byte aligned, and a cache so large that 32-bit row offsets do not reach its tail (cap x D x 2 B x 3 heads >= 2 GiB) -
    and `update` then takes the stage-by-stage route.  That also holds for a caller that asked for the next layer's id shift
    (the Qwen2-VL patch always does): ids, scores and caches equal an aligned / normal-sized run bit for bit."""
    import retake.longvideo_cache as lc

    layers, L = 3, 640
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    g = torch.Generator(device=dev()).manual_seed(12)
    q, k, v = ((1.7 * torch.randn((1, h, L, D), generator=g, device=dev())).to(torch.bfloat16) for h in (Hq, Hkv, Hkv))
    buf = torch.empty(q.numel() + 8, dtype=torch.bfloat16, device=dev())
    q_off = buf[1:1 + q.numel()].view(q.shape)          # the same numbers, 2 bytes off a 16-byte boundary
    q_off.copy_(q)
    assert q_off.data_ptr() % 16 == 2

    def run(qq, reserve):
        cache = lc.PivotKVCache(cfg(layers), reserve_tokens=reserve)
        seen = []
        for c in range(3):
            pos = chunk_ids(c, L)
            cache.kvcache_compression, cache.keypatches_mask_chunk = True, None
            for l in range(layers):
                cache.shift_temporal_ids_(pos, l)
                seen.append(pos.clone())
                cache.update(k, v, l, {"query_states": qq, "position_ids": pos, "rotary_emb": rot, "mrope_section": SEC,
                                       "shift_next_position_ids": True})
            cache.after_forward()
        cache.check()
        return seen, [cache.key_cache[l].clone() for l in range(layers)], [cache.position_cache[l].clone() for l in range(layers)], \
            int(cache._batch.shift_ticket[0])

    seen_a, ka, pa, rode_a = run(q, 8 * L)
    seen_b, kb, pb, rode_b = run(q_off, 8 * L)
    big = (1 << 31) // (3 * D * 2) + 4096               # rows per head from which the KV heads' tails span 2 GiB
    seen_c, kc, pc, rode_c = run(q, big)
    assert rode_a >= layers - 1 and rode_b == 0 and rode_c == 0   # the declined calls never launched the fused kernel
    for seen, ks_, ps_ in ((seen_b, kb, pb), (seen_c, kc, pc)):
        assert all(torch.equal(x, y) for x, y in zip(seen_a, seen))
        for l in range(layers):
            assert torch.equal(ka[l], ks_[l]) and torch.equal(pa[l], ps_[l])


def test_in_launch_shift_words_serve_one_stream_at_a_time():
    """The arrival counters belong to the batch: two update launches of one cache on DIFFERENT streams must not share them
    in flight.  A caller that switches its current stream between two layers gets a device synchronisation first (as the
    in-place compaction's words do, `_order_compaction`), so the result equals the single-stream run and nothing latches."""
    import retake.longvideo_cache as lc

    layers, L = 4, 640
    rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
    g = torch.Generator(device=dev()).manual_seed(9)
    q, k, v = ((1.7 * torch.randn((1, h, L, D), generator=g, device=dev())).to(torch.bfloat16) for h in (Hq, Hkv, Hkv))
    side = torch.cuda.Stream(device=dev())

    def run(alternate):
        cache = lc.PivotKVCache(cfg(layers), reserve_tokens=8 * L)
        for c in range(3):
            pos = chunk_ids(c, L)
            cache.kvcache_compression, cache.keypatches_mask_chunk = True, None
            for l in range(layers):
                use = side if (alternate and l % 2) else torch.cuda.current_stream(dev())
                use.wait_stream(torch.cuda.current_stream(dev()))
                with torch.cuda.stream(use):
                    cache.shift_temporal_ids_(pos, l)
                    cache.update(k, v, l, {"query_states": q, "position_ids": pos, "rotary_emb": rot, "mrope_section": SEC,
                                           "shift_next_position_ids": True})
                torch.cuda.current_stream(dev()).wait_stream(use)
            cache.after_forward()
        cache.check()
        assert int(cache._batch.shift_ticket[31]) == 0 and int(cache._batch.shift_ticket[32:].abs().max()) == 0
        return [cache.key_cache[l].clone() for l in range(layers)], [cache.position_cache[l].clone() for l in range(layers)]

    ka, pa = run(False)
    kb, pb = run(True)
    for l in range(layers):
        assert torch.equal(ka[l], kb[l]) and torch.equal(pa[l], pb[l])


def test_update_and_shift_under_inference_mode():
    """torch.inference_mode() tensors keep no version counter (`._version` raises): the shift memo and the inv_freq stamp take
    them as "cannot tell" - every layer then launches its own (idempotent) shift - and the caches equal a no_grad run's."""
    import retake.longvideo_cache as lc

    layers, n_chunks, L = 3, 2, 640
    g = torch.Generator(device=dev()).manual_seed(8)
    q, k, v = ((1.7 * torch.randn((1, h, L, D), generator=g, device=dev())).to(torch.bfloat16) for h in (Hq, Hkv, Hkv))

    def run(ctx):
        with ctx():
            rot = synth.RotaryStub(synth.inv_freq(D), A, device=dev())
            cache = lc.PivotKVCache(cfg(layers), reserve_tokens=8 * L)
            for c in range(n_chunks):
                pos = chunk_ids(c, L) + 0   # (created inside the context)
                cache.kvcache_compression = True
                cache.keypatches_mask_chunk = None
                for l in range(layers):
                    cache.shift_temporal_ids_(pos, l)
                    cache.update(k, v, l, {"query_states": q, "position_ids": pos, "rotary_emb": rot, "mrope_section": SEC})
                cache.after_forward()
            return [cache.key_cache[l].clone() for l in range(layers)], [cache.position_cache[l].clone() for l in range(layers)]

    ka, pa = run(torch.no_grad)
    kb, pb = run(torch.inference_mode)
    for l in range(layers):
        assert torch.equal(ka[l], kb[l]) and torch.equal(pa[l], pb[l])
