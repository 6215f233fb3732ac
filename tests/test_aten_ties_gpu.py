"""The selection kernels against torch.topk ON THE DEVICE, with exact ties at the k-th value.

The reference selects with ATen (`s.masked_fill_(mask, 1.); s.topk(keep).indices.sort().values`, longvideo_cache.py:272-277;
`dis.topk(k, sorted=False, dim=1)` + sort, visual_compression.py:134-135 / :167-168), so which of several tied tokens
survive is whatever ATen's topk does on the backend the reference is dropped into.  Probed on the MI355X
(tools/probe_aten_ties.py -> profiles/r14_aten_ties.txt): ATen-on-ROCm keeps the LOWEST indices of the tied run, for
fp32 / bf16 / fp16 inputs, one-block and multi-block slice sizes, 1-D and per-row.  These tests run the reference's
expressions with torch next to rtk_pivotkv_select / rtk_dpselect_select on the same inputs and require identical sets,
so a change of either side's tie rule fails here.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def _probe():
    import probe_aten_ties as pat

    return pat


@pytest.mark.parametrize("case", range(30))
def test_pivotkv_select_equals_aten_topk_on_device(case):
    pat = _probe()
    cases = list(pat.pivotkv_cases())
    assert len(cases) == 30
    name, s32, mask, keep, dt = cases[case]
    dev = torch.device("cuda:0")
    s = torch.from_numpy(s32).to(dev).to(dt)
    s.masked_fill_(torch.from_numpy(mask).to(dev), 1.)                    # longvideo_cache.py:274
    picked = s.topk(keep).indices.sort().values.cpu().numpy()            # :276-277
    seen = s.float().cpu().numpy()
    rep = pat.tie_report(seen, picked, keep)
    assert rep["all_above_picked"] and rep["chosen_tied"] == rep["need"], name
    # what ATen does on this device: the lowest indices of the tied run
    assert rep["rule"] in ("lowest-index-first", "no choice"), (name, rep)
    np.testing.assert_array_equal(picked, pat.canonical(seen, keep))
    # the library's selection, both kernels (one workgroup / chip-wide), on the scores ATen saw
    for chipwide in (False, True):
        got = pat.rtk_select(seen if dt is not torch.float32 else s32, mask, keep, chipwide)
        np.testing.assert_array_equal(got, picked, err_msg=f"{name} chipwide={chipwide}")


@pytest.mark.parametrize("T,N,tgt,sync", [(2048, 196, 512, False), (2048, 196, 1024, False), (2048, 196, 512, True),
                                          (256, 144, 128, False), (64, 16, 16, False), (2048, 729, 204, False),
                                          (20, 7, 5, False), (20, 7, 5, True)])
def test_dpselect_select_equals_aten_topk_on_device(T, N, tgt, sync):
    """Plateau distances (six distinct values): ties among peaks, among non-peaks and across the t-th key of most rows."""
    import retake._native as nv

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(T * 31 + N * 7 + tgt + int(sync))
    dis = rng.integers(0, 6, size=(T, N)).astype(np.float32) / 8.0
    dis[0] = 1.0
    d = torch.from_numpy(dis).to(dev)
    idx = torch.empty((tgt,) if sync else (tgt, N), dtype=torch.int64, device=dev)
    mk = torch.empty((tgt, N), dtype=torch.bool, device=dev)
    keys = torch.empty((2, T) if sync else (N, T), dtype=torch.float32, device=dev)
    nv.check(nv.lib.rtk_dpselect_select(nv.ptr(d), T, N, tgt, 3, int(sync), nv.ptr(idx), nv.ptr(mk), nv.ptr(keys), nv.stream()),
             "rtk_dpselect_select")
    # the reference's own expressions on the device, from the distances (visual_compression.py:108-135 / :142-169)
    dd = d.mean(1) if sync else d.transpose(0, 1).contiguous()            # [T] / [N, T]
    rows = dd.unsqueeze(0) if sync else dd
    pad = torch.nn.functional.pad(rows, (1, 1), value=float("-inf"))
    peak = (rows > pad[:, :-2]) & (rows >= pad[:, 2:])                    # argrelmax
    k_ref = rows + 2.0 * peak
    kk = keys[0:1] if sync else keys
    if sync:   # the patch mean's summation order is the kernel's own: same keys to rounding, topk runs on the kernel's
        assert torch.allclose(kk, k_ref, atol=1e-6, rtol=0)
        k_ref = kk
    else:
        assert torch.equal(kk, k_ref)
    _, ti = k_ref.topk(tgt, sorted=False, dim=1)
    ti = ti.sort(dim=1).values
    want = ti[0] if sync else ti.transpose(0, 1)
    assert torch.equal(idx, want)
    n_tied_rows = 0
    kn, pn = k_ref.cpu().numpy(), ti.cpu().numpy()
    pat = _probe()
    for r in range(kn.shape[0]):
        rule = pat.tie_report(kn[r], pn[r], tgt)["rule"]
        assert rule in ("lowest-index-first", "no choice")
        n_tied_rows += rule == "lowest-index-first"
    assert n_tied_rows >= 1 or T == 20, "no row had a tie at its t-th key: the case proves nothing"
