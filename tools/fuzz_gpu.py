#!/usr/bin/env python3
"""Randomised differential run of PivotKVCache on the GPU (not a pytest: minutes, not seconds).

For random shapes / dtypes / masks / ratios / RoPE flavours it compresses the same chunks three ways and compares:
  A  the cache as shipped (live-key pass 2, batched flush where the shape allows)
  B  the cache with `skip_masked_columns=False` (the full pass 2)          -> caches, scores, kept sets BITWISE equal to A
  C  the CPU oracle (fp32 runs only)                                        -> scores <= 5e-6, kept sets margin-aware,
                                                                               kept V exact, kept K <= 1e-5, ids exact
B also draws `one_call_update` at random (the stage-by-stage route against rtk_pivotkv_update / rtk_pivotkv_flush) and
`in_place_compaction` (the staged evict + place launches against rtk_pivotkv_compact_batched, which A always runs).
  P  the attention prologue (update_pre_rope on the pre-RoPE projections, where it applies) against its own oracle run
     on the tensors rotated at the continuity-shifted ids (fp32: same bars as C; 16-bit: ids and kept-row count)
    python tools/fuzz_gpu.py [--seconds 240] [--seed 0]
"""
import argparse
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import retake.longvideo_cache as lc
import synth
from oracle import oracle as orc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    n_cases = n_oracle = n_pre = 0
    worst_score = worst_k = 0.0
    while time.time() < t_end:
        Hkv = int(rng.choice([1, 2, 4, 8]))
        G = int(rng.choice([1, 4, 6, 7]))
        Hq = Hkv * G
        D = 128 if rng.uniform() < 0.8 else int(rng.choice([32, 64]))
        L = int(rng.choice([1, 33, 200, 511, 512, 640, 1000, 2304, 3000]))
        dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
        ratio = float(rng.choice([0.01, 0.1, 0.25, 0.5, 0.9, 1.0]))
        mrate = float(rng.choice([0.0, 0.0, 0.3, 0.6, 1.0]))
        reforge = bool(rng.uniform() < 0.8)
        mrope = bool(rng.uniform() < 0.6)
        layers = int(rng.integers(1, 4))
        chunks = int(rng.integers(1, 3))
        native = bool(rng.uniform() < 0.5)
        rounding = "fp32"
        if D == 128 and dtype == torch.bfloat16:
            rounding = str(rng.choice(["fp32", "fp32", "reference", "fast"]))
        elif D == 128 and dtype == torch.float16:
            rounding = str(rng.choice(["fp32", "fp32", "reference"]))
        sec = None
        if mrope:
            h = D // 2
            s1 = h // 4
            sec = [s1, (h - s1) // 2, h - s1 - (h - s1) // 2]
        S = synth.YARN_FACTOR4_ATTENTION_SCALING if rng.uniform() < 0.7 else 1.0
        rot = synth.RotaryStub(synth.inv_freq(D), S, device=dev)
        rot_cpu = synth.RotaryStub(synth.inv_freq(D), S)
        desc = (f"Hq={Hq} Hkv={Hkv} D={D} L={L} {str(dtype)[6:]} ratio={ratio} mask={mrate} reforge={reforge} mrope={mrope} "
                f"layers={layers} chunks={chunks} native_rope={native} a={S:.3f} score_rounding={rounding}")

        staged_b = bool(rng.uniform() < 0.5)
        inplace_b = bool(rng.uniform() < 0.5)
        fel_b = int(rng.choice([0, 0, 1, 2]))            # twin flushed every N layers (slot = layer mod N)
        ops_p = str(rng.choice(["reference", "pre_rope"]))   # what the prologue cache scores
        shn_b = bool(rng.uniform() < 0.5)                # twin: the next layer's id shift rides in the update launch
        desc += (f" staged_twin={staged_b} twin_in_place_compaction={inplace_b} twin_flush_every_layers={fel_b} "
                 f"twin_shift_next={shn_b} prologue_operands={ops_p}")

        def make(skip, **extra):
            kw = {"compression_ratio": ratio, "compression_method": "pivotkv", "pos_embed_reforge": reforge,
                  "native_rope": native, "skip_masked_columns": skip,
                  # ratio 1: `ca` keeps the chunk without scoring it, `cb` scores like the reference
                  "score_when_keeping_all": not skip, "score_rounding": rounding}
            kw.update(extra)
            cfg = types.SimpleNamespace(hidden_size=Hq * D, num_hidden_layers=layers, num_attention_heads=Hq,
                                        num_key_value_heads=Hkv,
                                        longvideo_kwargs={"kvcache_compression": True, "kvcache_compression_kwargs": kw})
            return lc.build_kvcache(cfg)

        ca, cb = make(True), make(False, one_call_update=not staged_b, in_place_compaction=inplace_b, flush_every_layers=fel_b,
                                    shift_next_in_update=shn_b)
        # the prologue route, where it applies (reforging cache, inv_freq rotary, chunks of >= 512 tokens; the reference's
        # rounding chain needs the reference's operands)
        pre = reforge and native and L >= 512 and (rounding != "reference" or ops_p == "reference")
        cp = make(True, score_when_keeping_all=True, prologue_operands=ops_p) if pre else None
        ocp = [orc.OraclePivotKV(Hq, Hkv, D, ratio, reforge) for _ in range(layers)] if pre and dtype == torch.float32 and L <= 1000 else None
        oc = [orc.OraclePivotKV(Hq, Hkv, D, ratio, reforge) for _ in range(layers)] if dtype == torch.float32 and L <= 1000 else None
        keep = max(1, int(ratio * L))
        try:
            for c in range(chunks):
                seed = int(rng.integers(0, 1 << 30))
                mask_np = rng.uniform(size=L) < mrate
                mask = torch.from_numpy(mask_np).to(dev) if mrate > 0 or rng.uniform() < 0.5 else None
                if mrope:
                    gw = max(1, int(np.sqrt(L)))
                    t = (np.arange(L) // max(1, gw * 2)) + 7 * c + 3
                    pos_np = np.stack([t, (np.arange(L) // gw) % 7 + 2, np.arange(L) % gw + 2]).reshape(3, 1, L).astype(np.int64)
                else:
                    pos_np = (np.arange(L) + 11 * c + 5).reshape(1, L).astype(np.int64)
                for l in range(layers):
                    q0, k0, v = synth.qkv_chunk(seed + l, Hq, Hkv, L, D)
                    pos_t = torch.from_numpy(pos_np)
                    q = synth.rope_forward(torch.from_numpy(q0), pos_t, rot_cpu, sec).to(dtype)
                    k = synth.rope_forward(torch.from_numpy(k0), pos_t, rot_cpu, sec).to(dtype)
                    vt = torch.from_numpy(v).to(dtype)
                    for cache in (ca, cb):
                        cache.keypatches_mask_chunk = mask
                        cache.kvcache_compression = True
                        # (the Qwen2-VL patch's opt-in: the launch may shift the ids it is handed - a clone here - for the next
                        # layer; the twin's `shift_next_in_update` draws the kill switch, so both forms of the launch run)
                        kw = {"query_states": q.to(dev), "position_ids": pos_t.to(dev).clone(), "rotary_emb": rot,
                              "shift_next_position_ids": True}
                        if sec:
                            kw["mrope_section"] = list(sec)
                        cache.update(k.to(dev), vt.to(dev), l, kw)
                    if oc is not None:
                        oc[l].keypatches_mask_chunk = mask_np if mask is not None else None
                        oc[l].update(k.numpy(), v, 0, q=q.numpy(), position_ids=pos_np, rotary=rot_cpu, mrope_section=sec)
                    if cp is not None:
                        cp.keypatches_mask_chunk = mask
                        cp.kvcache_compression = True
                        # projections in the [1, L, H*D] layout; the ids tensor is shared by the layers of the chunk
                        qd = torch.from_numpy(q0).to(dtype).to(dev).transpose(1, 2).contiguous().transpose(1, 2)
                        kd = torch.from_numpy(k0).to(dtype).to(dev).transpose(1, 2).contiguous().transpose(1, 2)
                        if l == 0:
                            pos_dev = torch.from_numpy(pos_np).to(dev)
                        got = cp.update_pre_rope(qd, kd, vt.to(dev), l, pos_dev, rot, list(sec) if sec else None,
                                                 query_out=[None, qd, torch.empty_like(qd)][int(rng.integers(0, 3))])
                        assert got is not None, "the prologue declined a chunk it should serve"
                        if ocp is not None:
                            prev = ocp[l].get_prev_temporal_idx(0)
                            psh = pos_np.copy()
                            if mrope:
                                psh[0, 0] += prev + 1 - psh[0, 0, 0]
                            else:
                                psh[0] += prev + 1 - psh[0, 0]
                            pst = torch.from_numpy(psh)
                            qs = synth.rope_forward(torch.from_numpy(q0), pst, rot_cpu, sec)
                            ks = synth.rope_forward(torch.from_numpy(k0), pst, rot_cpu, sec)
                            ocp[l].keypatches_mask_chunk = mask_np if mask is not None else None
                            ocp[l].update(ks.numpy(), v, 0, q=qs.numpy(), position_ids=psh, rotary=rot_cpu, mrope_section=sec)
                ca.after_forward()
                cb.after_forward()
                if cp is not None:
                    cp.after_forward()
                    for l in range(layers):
                        assert cp.key_cache[l].shape[2] == (c + 1) * keep
                        if ocp is not None:
                            last = ocp[l].last
                            so = last["score"]
                            sp = cp._batch.score[l].cpu().numpy()
                            bar = 2e-5 * max(1.0, float(np.abs(so).max()))
                            assert float(np.abs(sp - so).max()) < bar, "prologue route: score vs oracle"
                            idx = cp._batch.keep_idx[l].cpu().numpy()
                            xor = np.setxor1d(idx, last["keep_idx"])
                            if xor.size:
                                thr = np.sort(so)[::-1][keep - 1]
                                assert np.abs(so[xor] - thr).max() < 4 * bar, "prologue route: kept sets differ beyond noise"
                            else:
                                n0 = cp.key_cache[l].shape[2] - keep
                                assert np.array_equal(cp.value_cache[l][0, :, n0:].cpu().numpy(), last["kept_v"][0])
                                kerr = float(np.abs(cp.key_cache[l][0, :, n0:].cpu().numpy() - last["kept_k"][0]).max())
                                assert kerr <= 1e-5 * max(1.0, float(np.abs(last["kept_k"]).max())), f"prologue kept K {kerr}"
                                assert np.array_equal(cp.position_cache[l].cpu().numpy()[..., n0:].reshape(-1, keep),
                                                      last["pos"].reshape(-1, keep))
                            n_pre += 1
                for l in range(layers):
                    # (a twin flushed every N layers reuses its slots: its per-layer scores are gone, its caches are not)
                    sb = cb._batch.score[l] if not fel_b else None
                    sa = ca._batch.score[l] if ca._batch.score is not None else sb   # (keep-all batches allocate no scores)
                    if ca._batch.keep_all:
                        assert keep == L and ca.last_scores is None
                        sa = sb   # nothing was scored on the default route; the oracle checks the scored twin
                    if sb is not None:
                        assert torch.equal(sa, sb), "scores (after the mask override) differ between live-key and full pass 2"
                        assert torch.equal(ca._batch.keep_idx[l], cb._batch.keep_idx[l]), "kept sets differ"
                    assert torch.equal(ca.key_cache[l], cb.key_cache[l]) and torch.equal(ca.value_cache[l], cb.value_cache[l])
                    if reforge:
                        assert torch.equal(ca.position_cache[l], cb.position_cache[l])
                    if oc is not None and sa is not None:
                        last = oc[l].last
                        so = last["score"]
                        err = float(np.abs(sa.cpu().numpy() - so).max())
                        worst_score = max(worst_score, err)
                        # the oracle takes the module's tables (torch's libm cos / sin); native_rope computes them
                        # correctly rounded: one fp32 ulp apart in ~5 % of the entries
                        bar = (2e-5 if native and reforge else 5e-6) * max(1.0, float(np.abs(so).max()))
                        assert err < bar, f"score vs oracle {err} (bar {bar})"
                        idx = ca._batch.keep_idx[l].cpu().numpy()
                        xor = np.setxor1d(idx, last["keep_idx"])
                        if xor.size:
                            thr = np.sort(so)[::-1][keep - 1]
                            assert np.abs(so[xor] - thr).max() < 4 * bar, "kept sets differ beyond fp32 noise at the threshold"
                        else:
                            n0 = ca.key_cache[l].shape[2] - keep
                            assert np.array_equal(ca.value_cache[l][0, :, n0:].cpu().numpy(), last["kept_v"][0])
                            kerr = float(np.abs(ca.key_cache[l][0, :, n0:].cpu().numpy() - last["kept_k"][0]).max())
                            worst_k = max(worst_k, kerr)
                            assert kerr <= 1e-5 * max(1.0, float(np.abs(last["kept_k"]).max())), f"kept K vs oracle {kerr}"
                            if reforge:
                                assert np.array_equal(ca.position_cache[l].cpu().numpy()[..., n0:].reshape(-1, keep),
                                                      last["pos"].reshape(-1, keep))
                        n_oracle += 1
        except Exception as e:   # noqa: BLE001
            print("FUZZ FAILURE:", desc, "->", type(e).__name__, e, flush=True)
            raise
        n_cases += 1
    print(f"fuzz ok: {n_cases} random configurations (live-key == full pass 2 and one-call == staged route, bitwise), {n_oracle} "
          f"layer-chunks against the CPU oracle (max score diff {worst_score:.2e}, max kept-K diff {worst_k:.2e}), {n_pre} "
          f"prologue layer-chunks against the oracle", flush=True)
    fuzz_dpselect(dev, rng, a.seconds / 4)


def fuzz_dpselect(dev, rng, seconds):
    """DPSelect on random shapes / dtypes against the CPU oracle, row by row (golden_util.check_dpselect_bf16 with the
    oracle's result in the role of the reference's)."""
    import golden_util as gu
    import retake.visual_compression as vc

    t_end = time.time() + seconds
    n = relaxed = 0
    while time.time() < t_end:
        T = int(rng.choice([2, 3, 9, 33, 64, 130, 300]))
        N = int(rng.choice([2, 3, 7, 16, 50]))
        C = int(rng.choice([8, 24, 64, 200, 1280, 2056]))
        sync = bool(rng.uniform() < 0.4)
        tgt = int(rng.integers(1, T + 1))
        dt = ["fp32", "bf16", "fp16"][int(rng.integers(0, 3))]
        x = synth.frames_video(int(rng.integers(0, 1 << 30)), T, N, C)
        xt = torch.from_numpy(x)
        if dt == "fp32":
            xo, xd = x, xt.to(dev)
        elif dt == "bf16":
            xb = xt.bfloat16()
            xo, xd = xb.view(torch.int16).numpy().view(np.uint16), xb.to(dev)
        else:
            xh = xt.half()
            xo, xd = xh.view(torch.int16).numpy().view(np.float16), xh.to(dev)
        desc = f"DPSelect T={T} N={N} C={C} {dt} sync={sync} tgt={tgt}"
        try:
            out, mask, idx, dis, _ = vc.dpselect_stages(xd, tgt, 3, sync)
            o_out, o_mask, o_idx, o_dis = orc.dpselect(xo, tgt, 3, sync)
            g = {"sync": sync, "tgt": tgt, "dis32": o_dis, "idx": o_idx, "mask": o_mask}
            d = np.abs(dis.cpu().numpy() - o_dis)
            assert d.max() <= (2e-6 if dt == "fp32" else 2 ** -7), d.max()
            if dt == "fp32":   # fp32 distances differ by summation order only: treat them as perturbed within that noise
                st = gu.check_dpselect_bf16(g, dis.cpu().numpy(), idx.cpu().numpy(), mask.flatten().cpu().numpy())
            else:
                st = gu.check_dpselect_bf16(g, dis.cpu().numpy(), idx.cpu().numpy(), mask.flatten().cpu().numpy())
            relaxed += st["relaxed"]
            ii = idx if not sync else idx[:, None].expand(-1, N)
            assert torch.equal(out[0], torch.gather(xd[0], 0, ii[:, :, None].expand(-1, -1, C)))
        except Exception as e:   # noqa: BLE001
            print("FUZZ FAILURE:", desc, "->", type(e).__name__, e, flush=True)
            raise
        n += 1
    print(f"fuzz ok: {n} random DPSelect calls against the CPU oracle ({relaxed} rows took the relaxed rule)", flush=True)


if __name__ == "__main__":
    main()
