"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/retake_hip.h declares; the Python surface keeps the reference's names; the product
refuses to run without ROCm tensors (no CPU fallback)."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "retake_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import retake._native as nv

    lib = ctypes.CDLL(nv.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in retake_hip.h but not exported"
    assert set(syms) == set(nv.EXPORTS), "ctypes signature table out of sync with the header"
    assert nv.lib.rtk_version() == nv.ABI_VERSION
    assert nv.lib.rtk_arch() == b"gfx950"


def test_argument_validation_without_gpu():
    import retake._native as nv

    # NULL pointers / bad sizes are rejected on the host before any launch
    assert nv.lib.rtk_dpselect_dis(None, 4, 4, 4, 0, None, None) == nv.RTK_EINVAL
    assert b"NULL" in nv.lib.rtk_last_error()
    assert nv.lib.rtk_pivotkv_select(None, None, 8, 9, None, 0, 0, None, None, None, 0, None, 0, None) == nv.RTK_EINVAL
    assert nv.lib.rtk_pivotkv_score_workspace_bytes(28, 4, 6272, 128, 1) > 28 * 6272 * 128 * 2
    with pytest.raises(ValueError):
        nv.check(nv.RTK_EINVAL, "x")
    with pytest.raises(IndexError):
        nv.check(nv.RTK_EREFCRASH, "x")
    with pytest.raises(NotImplementedError):
        nv.check(nv.RTK_EUNSUPPORTED, "x")


def test_reference_surface_names_and_signatures():
    import retake.longvideo_cache as lc
    import retake.visual_compression as vc

    sig = inspect.signature(vc.memory_bank_compress_keyframe)
    assert list(sig.parameters) == ["memory_bank", "tgt_mem_len", "window_size", "sync"]
    assert sig.parameters["window_size"].default == 3 and sig.parameters["sync"].default is True
    for name in ("repeat_kv", "rotate_half", "apply_multimodal_rotary_pos_emb", "apply_rotary_pos_emb", "PivotKVCache",
                 "build_kvcache"):
        assert hasattr(lc, name)
    sig = inspect.signature(lc.PivotKVCache.update)
    assert list(sig.parameters) == ["self", "key_states", "value_states", "layer_idx", "cache_kwargs"]
    for m in ("before_forward", "after_forward", "update_num_evicted_tokens", "update_position_ids",
              "get_prev_temporal_idx"):
        assert callable(getattr(lc.PivotKVCache, m))


def test_no_cpu_fallback():
    import retake.visual_compression as vc

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vc.memory_bank_compress_keyframe(torch.randn(1, 8, 4, 16), 4, 3, True)


def test_degenerate_calls_raise_the_reference_exception_classes():
    """The reference's own failures on degenerate DPSelect / MA-LLM calls (recorded from the imported reference by
    tests/golden/gen_golden.py::gen_dpselect_degenerate; visual_compression.py:19-23, :100-123, :134, :153-156, :167):
    the binding raises the same exception class, from the argument checks, before any device work."""
    import numpy as np
    import retake.visual_compression as vc

    g = np.load(os.path.join(ROOT, "tests", "golden", "dpselect_edge_degenerate.npz"), allow_pickle=False)
    excs = {"RuntimeError": RuntimeError, "IndexError": IndexError}
    x1, x = torch.randn(1, 1, 4, 8), torch.randn(1, 6, 4, 8)
    calls = {
        "keyframe_T1_sync": lambda: vc.memory_bank_compress_keyframe(x1, 1, 3, True),
        "keyframe_T1_async": lambda: vc.memory_bank_compress_keyframe(x1, 1, 3, False),
        "keyframe_T1_N1_async": lambda: vc.memory_bank_compress_keyframe(x1[:, :, :1], 1, 3, False),
        "keyframe_tgt_gt_T_sync": lambda: vc.memory_bank_compress_keyframe(x, 7, 3, True),
        "keyframe_tgt_gt_T_async": lambda: vc.memory_bank_compress_keyframe(x, 7, 3, False),
        "keyframe_tgt_neg_sync": lambda: vc.memory_bank_compress_keyframe(x, -1, 3, True),
        "keyframe_tgt_gt_T_N1_async": lambda: vc.memory_bank_compress_keyframe(x[:, :, :1], 7, 3, False),
        "mallm_T1": lambda: vc.memory_bank_compress_MALLM(x1, torch.ones(1, 1, 4)),
        "mallm_hard_T1": lambda: vc.memory_bank_compress_MALLM_hard(x1),
    }
    for name, call in calls.items():
        want = excs[str(g[name])]
        with pytest.raises(want) as ei:
            call()
        assert type(ei.value) is want, (name, type(ei.value))   # not a subclass such as the no-CPU-fallback error's
        assert "no CPU fallback" not in str(ei.value), name


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "video-retake_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cuh", ".h", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "oracle" not in txt.lower(), f"{fn} mentions the oracle"


def test_build_kvcache_dispatch():
    import types

    import retake.longvideo_cache as lc

    cfg = types.SimpleNamespace()
    assert type(lc.build_kvcache(cfg)) is lc.DynamicCache
    cfg.longvideo_kwargs = {"kvcache_compression": False}
    assert type(lc.build_kvcache(cfg)) is lc.DynamicCache
    cfg = types.SimpleNamespace(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                                longvideo_kwargs={"kvcache_compression": True,
                                                  "kvcache_compression_kwargs": {"compression_ratio": 0.5,
                                                                                 "compression_method": "PivotKV"}})
    c = lc.build_kvcache(cfg)
    assert isinstance(c, lc.PivotKVCache) and c.pos_embed_reforge is False and c.head_dim == 16
    assert c.get_prev_temporal_idx(0) == -1 and c.get_seq_length() == 0
    cfg.longvideo_kwargs["kvcache_compression_kwargs"]["compression_method"] = "snapkv"
    with pytest.raises(NotImplementedError):
        lc.build_kvcache(cfg)


def test_rope_helpers_match_formula():
    import retake.longvideo_cache as lc

    torch.manual_seed(0)
    q, k = torch.randn(1, 4, 6, 8), torch.randn(1, 2, 6, 8)
    cos, sin = torch.randn(1, 6, 8), torch.randn(1, 6, 8)
    a = 1.1386
    qe, ke = lc.apply_rotary_pos_emb(q, k, cos * a, sin * a)
    qr, kr = lc.apply_rotary_pos_emb(qe, ke, cos * a, sin * a, reverse=True, attention_scaling=a)
    # forward then reverse with a-scaled tables returns (cos^2+sin^2) x; with true cos/sin it is the identity
    th = torch.rand(1, 6, 4)
    c = torch.cat([th.cos(), th.cos()], -1)
    s = torch.cat([th.sin(), th.sin()], -1)
    qe, ke = lc.apply_rotary_pos_emb(q, k, c * a, s * a)
    qr, kr = lc.apply_rotary_pos_emb(qe, ke, c * a, s * a, reverse=True, attention_scaling=a)
    assert torch.allclose(qr, q, atol=1e-5) and torch.allclose(kr, k, atol=1e-5)
    assert lc.repeat_kv(k, 2).shape == (1, 4, 6, 8) and torch.equal(lc.repeat_kv(k, 2)[:, 1], k[:, 0])
    c3, s3 = torch.randn(3, 1, 6, 8), torch.randn(3, 1, 6, 8)
    qm, _ = lc.apply_multimodal_rotary_pos_emb(q, k, c3, s3, [1, 2, 1])
    sel = [0, 1, 1, 2, 0, 1, 1, 2]
    cm = torch.stack([c3[sel[d], 0, :, d] for d in range(8)], -1)[None]
    sm = torch.stack([s3[sel[d], 0, :, d] for d in range(8)], -1)[None]
    assert torch.allclose(qm, q * cm + lc.rotate_half(q) * sm)


def test_host_argument_blocks_match_the_header(tmp_path):
    """rtk_pivotkv_batch / rtk_layer_state / rtk_update_io and the unit structs are HOST memory the Python side fills:
    the ctypes mirrors in retake/_native.py must have the layout gcc gives the header's structs (sizes and the offsets of
    every field)."""
    import subprocess

    import retake._native as nv

    mirrors = {"rtk_layer_state": nv.LayerState, "rtk_pivotkv_batch": nv.PivotKVBatch, "rtk_update_io": nv.UpdateIO,
               "rtk_evict_unit": nv.EvictUnit, "rtk_select_unit": nv.SelectUnit, "rtk_place_unit": nv.PlaceUnit,
               "rtk_copy_unit": nv.CopyUnit, "rtk_p2p_peers": nv.P2PPeers}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "retake_hip.h"', "int main(void) {"]
    for cname, st in mirrors.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, st in mirrors.items():
        assert int(got[cname]) == ctypes.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(st, fname).offset, f"{cname}.{fname}"


def test_flag_and_code_constants_match_the_header(tmp_path):
    """The enum values Python passes through the ABI (dtype codes, update flags, compaction modes, error codes) are the
    header's."""
    import subprocess

    import retake._native as nv

    names = {"RTK_F32": nv.RTK_F32, "RTK_BF16": nv.RTK_BF16, "RTK_BF16_REFROUND": nv.RTK_BF16_REFROUND, "RTK_BF16_FAST": nv.RTK_BF16_FAST,
             "RTK_F16": nv.RTK_F16, "RTK_F16_REFROUND": nv.RTK_F16_REFROUND, "RTK_SCORE_MANY_UNITS": nv.RTK_SCORE_MANY_UNITS,
             "RTK_PREPARE_K_ONLY": nv.RTK_PREPARE_K_ONLY, "RTK_UPDATE_PRE_ROPE": nv.RTK_UPDATE_PRE_ROPE,
             "RTK_UPDATE_Q_IN_PLACE": nv.RTK_UPDATE_Q_IN_PLACE, "RTK_UPDATE_ROUNDTRIP": nv.RTK_UPDATE_ROUNDTRIP,
             "RTK_UPDATE_SHIFT_NEXT": nv.RTK_UPDATE_SHIFT_NEXT,
             "RTK_COMPACT_K_ROTATE": nv.COMPACT_K_ROTATE, "RTK_COMPACT_K_COPY": nv.COMPACT_K_COPY,
             "RTK_COMPACT_K_INPLACE": nv.COMPACT_K_INPLACE, "RTK_EINVAL": nv.RTK_EINVAL, "RTK_EUNSUPPORTED": nv.RTK_EUNSUPPORTED,
             "RTK_EWORKSPACE": nv.RTK_EWORKSPACE, "RTK_EHIP": nv.RTK_EHIP, "RTK_EREFCRASH": nv.RTK_EREFCRASH,
             "RTK_SCORE_PREPARE": nv.SCORE_PREPARE, "RTK_SCORE_PASSES": nv.SCORE_PASSES, "RTK_SCORE_FINALIZE": nv.SCORE_FINALIZE,
             "RTK_P2P_MAX_RANKS": nv.P2P_MAX_RANKS}
    lines = ['#include <stdio.h>', '#include "retake_hip.h"', "int main(void) {"]
    lines += [f'  printf("{n} %d\\n", (int)({n}));' for n in names]
    lines += ["  return 0;", "}"]
    src = tmp_path / "consts.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "consts"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, v in names.items():
        assert int(got[n]) == v, n
    assert nv.lib.rtk_version() == nv.ABI_VERSION == 17


def test_round4_entry_points_validate_on_the_host():
    """rtk_pivotkv_update / _flush / rtk_mallm_hard_chain / rtk_rope_rotate_rows reject bad argument blocks before any
    launch (no GPU here), with a message."""
    import retake._native as nv

    lib = nv.lib
    assert lib.rtk_pivotkv_update(None, None, 0, None, None) == nv.RTK_EINVAL and b"batch" in lib.rtk_last_error()
    b = nv.PivotKVBatch()
    assert lib.rtk_pivotkv_update(ctypes.addressof(b), None, 0, None, None) == nv.RTK_EINVAL   # zero geometry
    b.Hq, b.Hkv, b.L, b.D, b.keep, b.P, b.slots, b.dtype = 28, 4, 640, 128, 160, 3, 2, nv.RTK_BF16
    dummy = (ctypes.c_char * 64)()
    p = ctypes.addressof(dummy)
    b.keep_idx = b.v_stage = b.score_ws = b.partials = b.score = b.sel_ws = p
    ls, io = nv.LayerState(), nv.UpdateIO()
    rc = lib.rtk_pivotkv_update(ctypes.addressof(b), ctypes.addressof(ls), 5, ctypes.addressof(io), None)
    assert rc == nv.RTK_EINVAL and b"slot" in lib.rtk_last_error()
    io.q = io.k = io.v = p
    rc = lib.rtk_pivotkv_update(ctypes.addressof(b), ctypes.addressof(ls), 0, ctypes.addressof(io), None)
    assert rc == nv.RTK_EINVAL and b"no room" in lib.rtk_last_error()          # an unallocated layer
    ls.k = ls.v = p
    ls.cap = 4096
    rc = lib.rtk_pivotkv_update(ctypes.addressof(b), ctypes.addressof(ls), 0, ctypes.addressof(io), None)
    assert rc == nv.RTK_EUNSUPPORTED and b"reforge" in lib.rtk_last_error()    # no reforge / rotary bound: the per-stage calls
    states = (ctypes.c_void_p * 1)(ctypes.addressof(ls))
    slots = (ctypes.c_int32 * 1)(0)
    assert lib.rtk_pivotkv_flush(ctypes.addressof(b), states, slots, 1, None) == nv.RTK_EINVAL   # nothing pending
    assert b"pending" in lib.rtk_last_error() or b"staging" in lib.rtk_last_error()
    assert lib.rtk_pivotkv_flush(ctypes.addressof(b), states, slots, 3, None) == nv.RTK_EINVAL   # more layers than slots
    assert lib.rtk_mallm_hard_chain(None, 8, 4, 64, nv.RTK_BF16, 4, 0, None, None, None) == nv.RTK_EINVAL
    assert lib.rtk_mallm_hard_chain(p, 8, 4, 64, nv.RTK_BF16, 9, 0, p, p, None) == nv.RTK_EINVAL and b"target" in lib.rtk_last_error()
    assert lib.rtk_rope_rotate_rows(None, 0, 0, 1, 1, 1, 128, nv.RTK_BF16, None, 0, 1, 3, None, 1.0, None, 0, 1, None) == nv.RTK_EINVAL
    assert lib.rtk_rope_rotate_rows(p, 0, 128, 1, 1, 4, 128, nv.RTK_BF16, p, 0, 2, 3, p, 1.0, None, 0, 1, None) == nv.RTK_EINVAL
    assert b"pos_stride_p" in lib.rtk_last_error()
    assert lib.rtk_profile_copy(p, p, 24, None) == nv.RTK_EINVAL
    # RTK_UPDATE_SHIFT_NEXT (ABI 16): one cache line for the launch count + one per arrival counter, whatever the geometry
    assert lib.rtk_pivotkv_shift_ticket_ints(6272, 128) == lib.rtk_pivotkv_shift_ticket_ints(640, 64) == 32 * 65


def test_compaction_entry_point_validates_on_the_host():
    """rtk_pivotkv_compact_batched / rtk_pivotkv_compact_sync_ints (ABI 14): sizes and argument errors without a launch."""
    import retake._native as nv

    lib = nv.lib
    # 28 units x one head group x (32 header ints + 49 row blocks, rounded up to 32): keep 1568, head_dim 128, bf16
    assert lib.rtk_pivotkv_compact_sync_ints(28, 4, 1568, 128, nv.RTK_BF16) == 28 * 96
    assert lib.rtk_pivotkv_compact_sync_ints(28, 8, 1568, 128, nv.RTK_BF16) == 28 * 2 * 96      # two head groups
    assert lib.rtk_pivotkv_compact_sync_ints(3, 4, 100, 128, nv.RTK_F32) == 3 * 64               # 16 rows per block in fp32
    assert lib.rtk_pivotkv_compact_sync_ints(1, 4, 100, 100, nv.RTK_BF16) == 0                   # unsupported head_dim
    dummy = (ctypes.c_char * 256)()
    p = ctypes.addressof(dummy)
    units = (nv.CompactUnit * 1)()
    call = lambda mode, sync, n, P=1: lib.rtk_pivotkv_compact_batched(units, 1, 4, 128, 4, P, nv.RTK_BF16, mode, p, 1.0,  # noqa: E731
                                                                     None, 0, 1, sync, n, None)
    assert call(1, p, 256) == nv.RTK_EINVAL and b"NULL pointer" in lib.rtk_last_error()          # empty unit
    units[0].k_tail = units[0].v_tail = units[0].keep_idx = p
    units[0].k_src = p
    assert call(1, p, 256) == nv.RTK_EINVAL and b"own" in lib.rtk_last_error()                   # k_src aliases the tail
    units[0].k_src = p + 128
    assert call(0, p, 256) == nv.RTK_EINVAL and b"pos_src" in lib.rtk_last_error()               # rotation without the new ids
    assert call(1, None, 256) == nv.RTK_EINVAL and b"sync" in lib.rtk_last_error()
    assert call(1, p, 8) == nv.RTK_EWORKSPACE
    assert call(9, p, 256) == nv.RTK_EINVAL
    assert call(1, p, 256, P=2) == nv.RTK_EINVAL
    assert lib.rtk_pivotkv_compact_batched(None, 0, 4, 128, 4, 1, nv.RTK_BF16, 1, p, 1.0, None, 0, 1, p, 256, None) == nv.RTK_EINVAL


def test_product_defaults_are_the_benched_configuration():
    """bench.py builds its cache from the reference's YAML keys only (configs/retake_demo.yaml:18-24 + the ratio its
    dynamic rule writes): no build-specific option is needed to get the path the headline times."""
    import sys

    sys.path.insert(0, ROOT)
    import bench

    kw = bench.cache_kwargs()
    assert set(kw) == {"dynamic_compression_ratio", "compression_method", "pos_embed_reforge", "max_input_length",
                       "compression_ratio"}
    import retake.longvideo_cache as lc

    c = lc.build_kvcache(bench.make_cache_config(2))
    assert c.native_rope is True and c.one_call_update is True and c.score_rounding == "fp32" and c.overlap_streams == 0
    assert c.defer_rerotation is False and c.score_queries_in_place is True and c.in_place_compaction is True
    assert c.skip_masked_columns is True and c.score_when_keeping_all is False
    assert c.prologue_operands == "reference" and c.flush_every_layers == 0    # round 5: the reference's operands; one flush per chunk
    assert c.shift_next_in_update is True    # ... and the next layer's id shift rides in the update launch
    import pytest

    for bad in ({"prologue_operands": "rotated"}, {"flush_every_layers": -1}, {"score_rounding": "exact"}):
        cfg = bench.make_cache_config(2)
        cfg.longvideo_kwargs["kvcache_compression_kwargs"].update(bad)
        with pytest.raises(ValueError):
            lc.build_kvcache(cfg)
    assert lc.build_kvcache(bench.make_cache_config(2)).memory_footprint()["total"] == 0   # nothing allocated before the first update

    class Dyn:   # a rotary module whose frequencies depend on the sequence length has to be CALLED
        inv_freq, attention_scaling, rope_type = torch.ones(4), 1.0, "dynamic"

    class Static:
        inv_freq, attention_scaling, rope_type = torch.ones(4), 1.0, "yarn"

    a, b2 = Static(), Static()
    ra, rb = c._rotary(a, torch.device("cpu")), c._rotary(b2, torch.device("cpu"))
    assert ra is not None and ra is rb                       # equal contents: one entry, one batch for all layers
    assert c._rotary(Dyn(), torch.device("cpu")) is None


def test_a_dropped_cache_is_released_without_the_garbage_collector():
    """The key_cache / value_cache list views hold a WEAK reference to their cache: dropping the cache frees it (and its
    gigabytes of device buffers) at once instead of whenever a reference cycle gets collected."""
    import gc
    import sys
    import weakref

    sys.path.insert(0, ROOT)
    import bench
    import retake.longvideo_cache as lc

    gc.disable()
    try:
        c = lc.build_kvcache(bench.make_cache_config(2))
        r, view = weakref.ref(c), c.key_cache
        assert len(view) == 0
        del c
        assert r() is None
        with pytest.raises(ReferenceError):
            len(view)
    finally:
        gc.enable()


def test_inference_mode_tensors_have_no_version_counter_and_are_served():
    """Tensors created under torch.inference_mode() raise on `._version`; the two places that consult the counter (the
    snapshot stamp of a rotary module's inv_freq, the memo of an id tensor the previous layer's launch has shifted) must
    take them as "cannot tell" instead of raising inside `update`."""
    import types

    import torch

    import retake.longvideo_cache as lc

    with torch.inference_mode():
        inv = torch.arange(8, dtype=torch.float32)
        ids = torch.arange(16).view(1, 16)
    with pytest.raises(RuntimeError):
        inv._version
    assert lc._version_of(inv) is None and lc._version_of(ids) is None and lc._version_of(torch.arange(3)) == 0
    stamp = lc._inv_stamp(types.SimpleNamespace(inv_freq=inv, attention_scaling=1.0))
    assert stamp is not None and stamp[1] is None
