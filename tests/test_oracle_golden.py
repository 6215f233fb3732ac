"""Pins the CPU oracle (oracle/) to the golden vectors produced by the Python reference.

Bar (BASELINE.json north_star): selected frame indices and kept KV indices bit-exact; compressed
values within 1e-5.  Fixtures are seeded to contain no fragile decision (margins stored in each
file); exact-tie fixtures are compared by the documented rule instead (count + score equality).
"""
import numpy as np
import pytest

import golden_util as gu
import synth
from oracle import oracle as orc

DP_FP32 = [n for n in gu.names("dpselect_") if "edge" not in n and "bf16" not in n and "fp16" not in n]
DP_BF16 = [n for n in gu.names("dpselect_") if "bf16" in n]
DP_FP16 = [n for n in gu.names("dpselect_") if "fp16" in n]
PK = [n for n in gu.names("pivotkv_") if not n.startswith(("pivotkv_bf16_", "pivotkv_fp16_", "pivotkv_prerope_"))]


@pytest.mark.parametrize("name", DP_FP32)
def test_dpselect_fp32_matches_reference(name):
    g = gu.load(name)
    x = gu.dpselect_input(g)
    out, mask, idx, dis = orc.dpselect(x, int(g["tgt"]), int(g["window"]), bool(g["sync"]))
    # distance: fp32 summation-order noise only
    assert np.abs(dis - g["dis32"]).max() < 2e-6
    assert np.abs(dis - g["dis64"]).max() < 2e-6
    # indices bit-exact, mask bit-exact, frames are pure copies (crc of the whole output)
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(mask, g["mask"])
    assert synth.checksum(out) == int(g["out_crc"])


@pytest.mark.parametrize("name", DP_BF16)
def test_dpselect_bf16_matches_reference(name):
    """Oracle vs the reference's bf16 run, row by row (golden_util.check_dpselect_bf16): rows whose distances equal the
    reference's must carry its indices and its mask; a flipped bf16 rounding may only move what it can reach."""
    g = gu.load(name)
    x = gu.dpselect_input(g)  # uint16 bf16 bits
    out, mask, idx, dis = orc.dpselect(x, int(g["tgt"]), int(g["window"]), bool(g["sync"]))
    # bf16 distance has bf16 resolution (2^-8 near 1); the emulated rounding chain reproduces the
    # reference up to one bf16 ulp of the cosine on rare elements
    d = np.abs(dis - g["dis32"])
    assert d.max() <= 2 ** -7 and (d > 0).mean() < 0.005
    st = gu.check_dpselect_bf16(g, dis, idx, mask)
    print(f"\n[{name}] rows {st['rows']}: exact {st['exact']}, tied-boundary {st['tied']}, relaxed {st['relaxed']} "
          f"({st['flipped_entries']} of {d.size} distances flipped, {st['peak_flags_differing']} peak flags and "
          f"{st['indices_differing']} picks differ from the reference's)")
    assert (st["exact"] + st["tied"] >= 0.9 * st["rows"]) or (bool(g["sync"]) and st["rows"] == 1)


@pytest.mark.parametrize("sync", [True, False])
@pytest.mark.parametrize("tgt", [5, 20])
def test_dpselect_plateau_ties(sync, tgt):
    """Hand-built exact ties in dis (repeated frames scaled by powers of two) and a zero vector.
    The peak rule (first index wins) is pinned exactly via the ratio-1.0 mask; at t=5 the k-th
    boundary may fall inside exact ties where torch's order is backend-specific:
    there the selected KEY multiset must match."""
    g = gu.load(f"dpselect_edge_plateau_{'sync' if sync else 'async'}_t{tgt}")
    x = g["x"]
    dis = orc.dpselect_dis(x[0])
    # cos of identical directions is 1 +- 1 ulp depending on summation order: pin dis to 1 ulp, then
    # pin the tie RULES of stencil and selection on the reference's own distance matrix.
    assert np.abs(dis - g["dis32"]).max() <= 2.5e-7
    assert dis[3, 1] == 1.0 and dis[4, 1] == 1.0  # zero vector -> cos 0
    dis = g["dis32"]
    idx, mask2d, keys = orc.dpselect_select(dis, tgt, 3, sync)
    if tgt == 20:
        np.testing.assert_array_equal(idx, g["idx"])
        np.testing.assert_array_equal(mask2d.reshape(-1), g["mask"])
        np.testing.assert_array_equal(orc.gather_frames(x[0], idx, sync)[None], g["out"])
        return
    ref_idx = g["idx"]
    rows = [(keys, idx, ref_idx)] if sync else [(keys[n], idx[:, n], ref_idx[:, n]) for n in range(3)]
    for krow, mine, theirs in rows:
        np.testing.assert_array_equal(np.sort(krow[mine]), np.sort(krow[theirs]))


def test_dpselect_async_n1_is_reference_crash():
    g = gu.load("dpselect_edge_n1_async")
    assert str(g["exception"]) == "IndexError"
    with pytest.raises(IndexError):
        orc.dpselect(np.zeros((1, 8, 1, 16), np.float32), 4, 3, sync=False)


def _run_pivotkv(g):
    Hq, Hkv, D = int(g["Hq"]), int(g["Hkv"]), int(g["D"])
    sec = [int(s) for s in g["mrope_section"]] or None
    rotary = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]))
    cache = orc.OraclePivotKV(Hq, Hkv, D, float(g["ratio"]), bool(g["reforge"]))
    layer = int(g["layer"])
    for c in range(int(g["n_chunks"])):
        q, k, v, pos, mask = gu.pivotkv_chunk_inputs(g, c)
        cache.keypatches_mask_chunk = mask
        prev_len = 0 if len(cache.key_cache) <= layer or len(cache.key_cache[layer]) == 0 \
            else cache.key_cache[layer].shape[2]
        kout, vout = cache.update(k, v, layer, q=q, position_ids=pos, rotary=rotary, mrope_section=sec)
        assert kout.shape == (1, Hkv, prev_len + k.shape[2], D)
        yield c, cache, k, v


@pytest.mark.parametrize("name", PK)
def test_pivotkv_matches_reference(name):
    g = gu.load(name)
    keep = int(g["keep"])
    tie = bool(g["tie_case"])
    for c, cache, k, v in _run_pivotkv(g):
        pre = f"c{c}_"
        last = cache.last
        # score: fp32 noise vs the reference's own fp32 score and the fp64 recomputation
        assert np.abs(last["score"] - g[pre + "score32"]).max() < 5e-6
        assert np.abs(last["score"] - g[pre + "score64"]).max() < 5e-6
        if tie:
            # boundary inside exact 1.0 ties: same score multiset, canonical rule = lowest index first
            s = g[pre + "score32"]
            np.testing.assert_array_equal(np.sort(s[last["keep_idx"]]), np.sort(s[g[pre + "keep_idx"]]))
            thr = np.sort(s)[::-1][keep - 1]
            ties = np.nonzero(last["score"] == thr)[0]
            picked = np.intersect1d(ties, last["keep_idx"])
            np.testing.assert_array_equal(picked, ties[: len(picked)])
            break  # later chunks depend on which tied tokens were kept
        np.testing.assert_array_equal(last["keep_idx"], g[pre + "keep_idx"])
        assert np.abs(last["kept_k"] - g[pre + "kept_k"]).max() <= 1e-5
        if bool(g["raw"]):
            np.testing.assert_array_equal(last["kept_v"], g[pre + "kept_v"])
        else:
            assert synth.checksum(last["kept_v"]) == int(g[pre + "kept_v_crc"])
        layer = int(g["layer"])
        assert cache.num_evicted_tokens[layer] == int(g[pre + "num_evicted"])
        if bool(g["reforge"]):
            np.testing.assert_array_equal(cache.position_cache[layer], g[pre + "position_cache"])
    if not tie:
        assert len(cache.position_cache) == int(g["position_cache_len"])
        np.testing.assert_array_equal(np.array(cache.num_evicted_tokens), g["num_evicted_list"])


# ---------------------------------------------------------------------------------------------------
# PivotKV in the production dtype: the reference run on a bf16 model (fixtures pivotkv_bf16_*)
# ---------------------------------------------------------------------------------------------------
def check_bf16_against_reference(g, c, score, keep_idx, kept_k_bits, pos_new, what, max_bad=None):
    """Shared by the oracle test (CPU) and the HIP reference-rounding test (GPU).  The reference's bf16 score chain
    (longvideo_cache.py:264-270) is reproduced up to the summation order inside ATen's bf16 gemm / sums, which may move
    an isolated entry by ONE bf16 ulp; torch.topk's pick among exact ties is backend-defined, so the
    kept sets are compared through the scores of the tokens they disagree on.  Returns (score mismatches, kept xor)."""
    pre = f"c{c}_"
    L, keep = int(g["L"]), int(g["keep"])
    ref = orc.bf16_bits_to_f32(g[pre + "score_bf16"])
    assert np.array_equal(orc.bf16_round(score), score), f"{what}: scores must be bf16 values"
    bad = np.nonzero(score != ref)[0]
    limit = max(2, L // 500) if max_bad is None else max_bad
    assert bad.size <= limit, f"{what}: {bad.size} of {L} scores differ from the reference's (bar {limit})"
    if bad.size:
        assert (np.abs(score[bad] - ref[bad]) <= gu.bf16_ulp(np.minimum(np.abs(score[bad]), np.abs(ref[bad])))).all()
    ref_idx = g[pre + "keep_idx"]
    thr = np.sort(ref)[::-1][keep - 1]                        # the reference's k-th largest score
    xor = np.setxor1d(keep_idx, ref_idx)
    # every token the two sides disagree on scores within one bf16 ulp of the threshold (ties, or a 1-ulp neighbour)
    assert (np.abs(ref[xor] - thr) <= gu.bf16_ulp(np.full(xor.size, thr))).all(), f"{what}: kept sets differ beyond ties"
    np.testing.assert_array_equal(np.sort(ref[keep_idx])[bad.size + 2:], np.sort(ref[ref_idx])[bad.size + 2:])
    # canonical tie rule on this side: lowest index first
    ties = np.nonzero(score == np.sort(score)[::-1][keep - 1])[0]
    picked = np.intersect1d(ties, keep_idx)
    np.testing.assert_array_equal(picked, ties[: picked.size])
    # kept keys: bit-exact bf16 re-rotation for every token both sides kept at the same new position
    ref_pos = g[pre + "position_cache"][..., -keep:].reshape(-1, keep)
    common, ia, ib = np.intersect1d(keep_idx, ref_idx, return_indices=True)
    same_pos = (pos_new.reshape(-1, keep)[:, ia] == ref_pos[:, ib]).all(0)
    assert same_pos.mean() > 0.9
    a = kept_k_bits.reshape(-1, keep, kept_k_bits.shape[-1])[:, ia[same_pos]]
    b = g[pre + "kept_k_bits"].reshape(-1, keep, kept_k_bits.shape[-1])[:, ib[same_pos]]
    return bad.size, xor.size, a, b


@pytest.mark.parametrize("name", gu.names("pivotkv_bf16_"))
def test_pivotkv_bf16_chain_matches_reference(name):
    g = gu.load(name)
    Hq, Hkv, D, L = (int(g[k]) for k in ("Hq", "Hkv", "D", "L"))
    if L > 2000 and orc.num_threads() < 4:
        pytest.skip("L = 6272 needs a few cores")
    sec = [int(x) for x in g["mrope_section"]]
    rot = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]))
    oc = orc.OraclePivotKV(Hq, Hkv, D, float(g["ratio"]), True, bf16=True)
    q, k, v, pos, mask = gu.pivotkv_bf16_chunk_inputs(g, 0)   # later chunks depend on which tied tokens were kept
    oc.keypatches_mask_chunk = mask
    oc.update(orc.bf16_bits_to_f32(k), orc.bf16_bits_to_f32(v), 0, q=orc.bf16_bits_to_f32(q), position_ids=pos,
              rotary=rot, mrope_section=sec)
    last = oc.last
    kk = (np.ascontiguousarray(last["kept_k"]).view(np.uint32) >> 16).astype(np.uint16)
    nbad, nxor, a, b = check_bf16_against_reference(g, 0, last["score"], last["keep_idx"], kk, last["pos"], "oracle")
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", gu.names("pivotkv_prerope_bf16_"))
def test_pivotkv_bf16_chain_from_pre_rope_projections_matches_reference(name):
    """Round-5 fixtures: the reference run on a bf16 model FROM the bf16 pre-RoPE projections (its rotation helper in bf16,
    then PivotKVCache.update).  The oracle is handed the rotated tensors - the rotation restated with torch's bf16 ops and
    crc-pinned to the tensors the reference's own helper produced - every chunk, both RoPE flavours: scores equal the
    reference's bf16 scores up to isolated 1-ulp entries, kept set equal up to exact ties, kept K bit-exact, ids exact."""
    g = gu.load(name)
    Hq, Hkv, D, L = (int(g[k]) for k in ("Hq", "Hkv", "D", "L"))
    if L > 2000 and orc.num_threads() < 4:
        pytest.skip("L = 6272 needs a few cores")
    sec = [int(x) for x in g["mrope_section"]] or None
    rot = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]))
    oc = orc.OraclePivotKV(Hq, Hkv, D, float(g["ratio"]), True, bf16=True)
    for c in range(int(g["n_chunks"])):
        q0, k0, v, pos_in, pos, mask = gu.pivotkv_prerope_chunk_inputs(g, c)
        q, k = gu.rotate_like_a_bf16_model(g, c, q0, k0)
        prev = oc.get_prev_temporal_idx(0)
        shifted = pos_in.copy()
        shifted[0, ..., :] += prev + 1 - shifted.reshape(-1)[0]          # the attention patch's continuity shift
        np.testing.assert_array_equal(shifted, pos)
        oc.keypatches_mask_chunk = mask
        oc.update(orc.bf16_bits_to_f32(k), orc.bf16_bits_to_f32(v), 0, q=orc.bf16_bits_to_f32(q), position_ids=pos,
                  rotary=rot, mrope_section=sec)
        last = oc.last
        kk = (np.ascontiguousarray(last["kept_k"]).view(np.uint32) >> 16).astype(np.uint16)
        nbad, nxor, a, b = check_bf16_against_reference(g, c, last["score"], last["keep_idx"], kk, last["pos"], "oracle")
        np.testing.assert_array_equal(a, b)
        if nxor:
            break   # later chunks depend on which tied tokens were kept


@pytest.mark.parametrize("name", gu.names("pivotkv_prerope_fp16_"))
def test_pivotkv_fp16_chain_from_pre_rope_projections_matches_reference(name):
    """The float16 twin of the fixtures above: the reference's whole attention-side chain on an fp16 model from the fp16
    pre-RoPE projections; the oracle's fp16 chain (numpy-float16 RoPE, score_rounding 'reference16') reproduces its scores
    up to isolated 1-ulp entries, kept sets up to threshold ties, kept keys bit for bit."""
    g = gu.load(name)
    Hq, Hkv, D, L, keep = (int(g[k]) for k in ("Hq", "Hkv", "D", "L", "keep"))
    sec = [int(x) for x in g["mrope_section"]] or None
    rot = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]))
    oc = orc.OraclePivotKV(Hq, Hkv, D, float(g["ratio"]), True, fp16=True, score_rounding="reference16")
    for c in range(int(g["n_chunks"])):
        pre = f"c{c}_"
        q0, k0, v, pos_in, pos, mask = gu.pivotkv_prerope_chunk_inputs(g, c)
        q, k = gu.rotate_like_a_bf16_model(g, c, q0, k0)
        f32 = lambda b: b.view(np.float16).astype(np.float32)   # noqa: E731
        oc.keypatches_mask_chunk = mask
        oc.update(f32(k), f32(v), 0, q=f32(q), position_ids=pos, rotary=rot, mrope_section=sec)
        last = oc.last
        ref = f32(g[pre + "score_bf16"])                        # (the key's name is historical: bits of the model dtype)
        score = last["score"]
        bad = np.nonzero(score != ref)[0]
        assert bad.size <= max(4, L // 200), bad.size
        assert (np.abs(score[bad] - ref[bad]) <= gu.fp16_ulp(np.minimum(np.abs(score[bad]), np.abs(ref[bad])))).all()
        thr = np.sort(ref)[::-1][keep - 1]
        xor = np.setxor1d(last["keep_idx"], g[pre + "keep_idx"])
        assert (np.abs(ref[xor] - thr) <= gu.fp16_ulp(np.full(xor.size, thr))).all()
        ref_pos = g[pre + "position_cache"][..., -keep:].reshape(-1, keep)
        common, ia, ib = np.intersect1d(last["keep_idx"], g[pre + "keep_idx"], return_indices=True)
        same = (last["pos"].reshape(-1, keep)[:, ia] == ref_pos[:, ib]).all(0)
        a = last["kept_k"].astype(np.float16).view(np.uint16).reshape(-1, keep, D)[:, ia[same]]
        b = g[pre + "kept_k_bits"].reshape(-1, keep, D)[:, ib[same]]
        np.testing.assert_array_equal(a, b)
        print(f"\n[{name} c{c}] oracle fp16 chain: {bad.size} of {L} scores off by one fp16 ulp, kept xor {xor.size}")
        if xor.size:
            break


def test_topk_ties_lowest_index_first():
    v = np.array([1, 3, 3, 2, 3, 0, 3], dtype=np.float32)
    import ctypes as C

    idx = np.empty(3, dtype=np.int64)
    rc = orc.lib().orc_topk_sorted(v.ctypes.data_as(C.c_void_p), len(v), 3, idx.ctypes.data_as(C.c_void_p))
    assert rc == 0
    np.testing.assert_array_equal(idx, [1, 2, 4])


def test_position_rescale_float32_truncation():
    # keep/k_len = 22/75 is not dyadic: float32 multiply then truncate
    L, keep = 75, 22
    pos = np.arange(L, dtype=np.int64)[None] * 3 + 11
    idx = np.sort(np.random.default_rng(0).choice(L, keep, replace=False)).astype(np.int64)
    out = orc.pivotkv_positions(pos, idx, True)
    t = pos[0, idx]
    want = t.min() + ((t - t.min()).astype(np.float32) * np.float32(keep / L)).astype(np.int64)
    np.testing.assert_array_equal(out[0], want)


# ---------------------------------------------------------------------------------------------------
# MA-LLM / MA-LLM-hard merges (visual_compression.py:5-83)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", gu.names("mallm_"))
def test_mallm_matches_reference(name):
    g = gu.load(name)
    x = g["x"][0]
    bank, sizes, steps = orc.mallm_compress(x, int(g["tgt"]), bool(g["sync"]), bool(g["hard"]))
    np.testing.assert_array_equal(steps, g["steps_idx"])            # the merged pair of every step, bit-exact
    ref = g["out"][0]
    if str(g["dtype"]) == "fp32":
        if bool(g["hard"]):
            np.testing.assert_array_equal(bank, ref)                # pure copies
        else:
            assert np.abs(bank - ref).max() <= 1e-5
            np.testing.assert_array_equal(sizes, g["size"][0])
    else:
        a = (bank.astype(np.uint32) << 16).view(np.float32)
        b = (ref.astype(np.uint32) << 16).view(np.float32)
        assert np.abs(a - b).max() <= 2 ** -6                       # at most a bf16 ulp on values of O(1)
        assert (a != b).mean() < 0.02


# ---------------------------------------------------------------------------------------------------
# fp16: the reference run on float16 tensors (fixtures of gen_golden.py --only fp16)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", DP_FP16)
def test_dpselect_fp16_matches_reference(name):
    """The numpy-float16 restatement of ATen's fp16 cosine chain against the reference's fp16 run, row by row (the same
    checker as bf16: rows whose distances equal the reference's must carry its indices and its mask)."""
    g = gu.load(name)
    x = gu.dpselect_input(g).view(np.float16)
    out, mask, idx, dis = orc.dpselect(x, int(g["tgt"]), int(g["window"]), bool(g["sync"]))
    d = np.abs(dis - g["dis32"])
    assert d.max() <= 2 ** -9 and (d > 0).mean() < 0.01      # one fp16 ulp of a cosine below 1, rare
    st = gu.check_dpselect_bf16(g, dis, idx, mask)
    print(f"\n[{name}] rows {st['rows']}: exact {st['exact']}, tied-boundary {st['tied']}, relaxed {st['relaxed']} "
          f"({st['flipped_entries']} of {d.size} distances flipped, {st['peak_flags_differing']} peak flags and "
          f"{st['indices_differing']} picks differ from the reference's)")
    assert (st["exact"] + st["tied"] >= 0.9 * st["rows"]) or (bool(g["sync"]) and st["rows"] == 1)


def check_fp16_against_reference(g, score, keep_idx, kept_k16, pos_new, k_unrot16, what):
    """Shared by the oracle test (CPU) and the HIP test (GPU).  On float16 tensors the reference rounds the logits, the
    probabilities and the sums to fp16 (longvideo_cache.py:264-270); the build scores with exact products and fp32
    accumulation (like its bf16 default), so: the score is the exact score of the reference's own un-rotated operands
    (<= 2e-5), every token the kept sets disagree on has a reference score within one fp16 ulp of its threshold, the
    un-rotated keys equal the reference's bit for bit, and so do the re-rotated kept keys of every token both sides
    kept at the same new position."""
    L, keep = int(g["L"]), int(g["keep"])
    mask = g["c0_mask"]
    s64 = g["c0_score64"].copy()
    s64[mask] = 1.0
    err = float(np.abs(score - s64).max())
    assert err < 2e-5, f"{what}: score differs from the exact score of the reference's operands by {err}"
    ref = g["c0_score_bits"].view(np.float16).astype(np.float32)
    ref_idx = g["c0_keep_idx"]
    thr = np.sort(ref)[::-1][keep - 1]
    xor = np.setxor1d(keep_idx, ref_idx)
    assert (np.abs(ref[xor] - thr) <= gu.fp16_ulp(np.full(xor.size, thr))).all(), f"{what}: kept sets differ beyond ties"
    assert xor.size <= max(4, L // 100)
    if k_unrot16 is not None:
        np.testing.assert_array_equal(np.asarray(k_unrot16).view(np.uint16).reshape(-1),
                                      g["c0_k_unrot_bits"].reshape(-1), err_msg=f"{what}: un-rotated keys")
    ref_pos = g["c0_position_cache"][..., -keep:].reshape(-1, keep)
    common, ia, ib = np.intersect1d(keep_idx, ref_idx, return_indices=True)
    same_pos = (pos_new.reshape(-1, keep)[:, ia] == ref_pos[:, ib]).all(0)
    assert same_pos.mean() > 0.9
    a = np.asarray(kept_k16).view(np.uint16).reshape(-1, keep, int(g["D"]))[:, ia[same_pos]]
    b = g["c0_kept_k_bits"].reshape(-1, keep, int(g["D"]))[:, ib[same_pos]]
    np.testing.assert_array_equal(a, b, err_msg=f"{what}: re-rotated kept keys")
    return xor.size, err


def check_fp16_refchain_against_reference(g, score, keep_idx, kept_k16, pos_new, what, max_bad):
    """Shared by the oracle test (CPU) and the HIP test (GPU) of the reference's fp16 score chain (longvideo_cache.py:264-270
    on a float16 model: logits, probabilities, per-head sums and both means rounded to fp16).  Like the bf16 twin: the
    scores equal the reference's fp16 scores except isolated entries by one fp16 ulp (summation order inside ATen's
    gemm / sum kernels), the kept sets agree up to tokens within one ulp of the threshold, kept keys bit-exact."""
    L, keep = int(g["L"]), int(g["keep"])
    ref = g["c0_score_bits"].view(np.float16).astype(np.float32)
    assert np.array_equal(score.astype(np.float16).astype(np.float32), score), f"{what}: scores must be fp16 values"
    bad = np.nonzero(score != ref)[0]
    assert bad.size <= max_bad, f"{what}: {bad.size} of {L} scores differ from the reference's (bar {max_bad})"
    if bad.size:
        assert (np.abs(score[bad] - ref[bad]) <= gu.fp16_ulp(np.minimum(np.abs(score[bad]), np.abs(ref[bad])))).all()
    ref_idx = g["c0_keep_idx"]
    thr = np.sort(ref)[::-1][keep - 1]
    xor = np.setxor1d(keep_idx, ref_idx)
    assert (np.abs(ref[xor] - thr) <= gu.fp16_ulp(np.full(xor.size, thr))).all(), f"{what}: kept sets differ beyond ties"
    ties = np.nonzero(score == np.sort(score)[::-1][keep - 1])[0]   # canonical tie rule on this side: lowest index first
    picked = np.intersect1d(ties, keep_idx)
    np.testing.assert_array_equal(picked, ties[: picked.size])
    ref_pos = g["c0_position_cache"][..., -keep:].reshape(-1, keep)
    common, ia, ib = np.intersect1d(keep_idx, ref_idx, return_indices=True)
    same_pos = (pos_new.reshape(-1, keep)[:, ia] == ref_pos[:, ib]).all(0)
    assert same_pos.mean() > 0.9
    a = np.asarray(kept_k16).view(np.uint16).reshape(-1, keep, int(g["D"]))[:, ia[same_pos]]
    b = g["c0_kept_k_bits"].reshape(-1, keep, int(g["D"]))[:, ib[same_pos]]
    np.testing.assert_array_equal(a, b, err_msg=f"{what}: re-rotated kept keys")
    return bad.size, xor.size


def test_oracle_fp16_rounding_is_numpy_float16():
    """gcc 11 has no _Float16 on x86-64: the C oracle spells the fp16 rounding out; here against numpy.float16 over normal,
    subnormal, overflowing and special values."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(50000).astype(np.float32) * s for s in (1e-8, 1e-6, 1e-4, 1e-2, 1, 100, 3e4, 7e4)]
                       + [np.array([65504, 65519.99, 65520, 65536, 6e-8, 2.98e-8, 2.99e-8, 0, -0.0, np.inf, -np.inf], dtype=np.float32)])
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).astype(np.float32)
    assert np.array_equal(ref.view(np.uint32), orc.round_fp16(x).view(np.uint32))


@pytest.mark.parametrize("name", gu.names("pivotkv_fp16_"))
def test_pivotkv_fp16_reference_chain_matches_reference(name):
    """The oracle's fp16 score chain (score_rounding 'reference16') against the reference's own fp16 run."""
    g = gu.load(name)
    Hq, Hkv, D, L = (int(g[k]) for k in ("Hq", "Hkv", "D", "L"))
    sec = [int(x) for x in g["mrope_section"]]
    rot = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]))
    oc = orc.OraclePivotKV(Hq, Hkv, D, float(g["ratio"]), True, fp16=True, score_rounding="reference16")
    q, k, v, pos, mask = gu.pivotkv_fp16_chunk_inputs(g)
    oc.keypatches_mask_chunk = mask
    oc.update(k.astype(np.float32), v.astype(np.float32), 0, q=q.astype(np.float32), position_ids=pos, rotary=rot,
              mrope_section=sec)
    last = oc.last
    nbad, nxor = check_fp16_refchain_against_reference(g, last["score"], last["keep_idx"], last["kept_k"].astype(np.float16),
                                                       last["pos"], "oracle fp16 chain", max_bad=max(2, L // 500))
    print(f"\n[{name}] oracle fp16 chain vs the reference's fp16 run: {nbad} of {L} scores differ by one fp16 ulp, kept xor {nxor}")


@pytest.mark.parametrize("name", gu.names("pivotkv_fp16_"))
def test_pivotkv_fp16_chain_matches_reference(name):
    g = gu.load(name)
    Hq, Hkv, D, L = (int(g[k]) for k in ("Hq", "Hkv", "D", "L"))
    sec = [int(x) for x in g["mrope_section"]]
    rot = synth.RotaryStub(g["inv_freq"], float(g["attention_scaling"]))
    oc = orc.OraclePivotKV(Hq, Hkv, D, float(g["ratio"]), True, fp16=True, score_rounding="fp32")
    q, k, v, pos, mask = gu.pivotkv_fp16_chunk_inputs(g)
    oc.keypatches_mask_chunk = mask
    oc.update(k.astype(np.float32), v.astype(np.float32), 0, q=q.astype(np.float32), position_ids=pos, rotary=rot,
              mrope_section=sec)
    last = oc.last
    nxor, err = check_fp16_against_reference(g, last["score"], last["keep_idx"], last["kept_k"].astype(np.float16),
                                             last["pos"], last["k_unrot"].astype(np.float16), "oracle")
    print(f"\n[{name}] oracle vs the reference's fp16 run: {nxor // 2} kept tokens differ (threshold ties), "
          f"max |score - exact| {err:.2e}")
    np.testing.assert_array_equal(last["kept_v"].astype(np.float16).view(np.uint16)[0], v.view(np.uint16)[0][:, last["keep_idx"]])
