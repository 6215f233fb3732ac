"""Which tied elements does torch.topk pick ON THIS DEVICE?  

The reference selects with ATen: `s.masked_fill_(mask, 1.); s.topk(keep).indices.sort().values`
(longvideo_cache.py:272-277) and `dis.topk(k, sorted=False, dim=1)` + `sort` (visual_compression.py:134-135,
:167-168).  Ties at the k-th value are resolved by whatever the backend's topk does; on the ROCm device the
reference is dropped into, that is ATen's radix select.  This probe runs exactly those expressions with torch on
the GPU over inputs with exact ties straddling the boundary and writes which tied elements were chosen, next to
what the library's own selection kernels return on the same inputs.

    python tools/probe_aten_ties.py [out.txt]        (needs the GPU; prints + writes the report)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import retake._native as nv  # noqa: E402

DEV = torch.device("cuda:0")


def classify(picked_tied_ranks: np.ndarray, n_tied: int, need: int) -> str:
    """picked_tied_ranks: ranks (0 = lowest index) among the tied elements that were chosen."""
    r = np.sort(picked_tied_ranks)
    if need == 0 or need == n_tied:
        return "no choice"
    if np.array_equal(r, np.arange(need)):
        return "lowest-index-first"
    if np.array_equal(r, np.arange(n_tied - need, n_tied)):
        return "highest-index-first"
    return "other"


def tie_report(values: np.ndarray, picked: np.ndarray, k: int):
    """values: what topk saw (after the mask override), picked: sorted indices it returned."""
    v = values.astype(np.float64)
    kth = np.sort(v)[::-1][k - 1]
    above = np.flatnonzero(v > kth)
    tied = np.flatnonzero(v == kth)
    need = k - above.size
    ok_above = np.isin(above, picked).all()
    chosen_tied = np.intersect1d(picked, tied)
    ranks = np.searchsorted(tied, chosen_tied)
    return {"kth": float(kth), "n_above": int(above.size), "n_tied": int(tied.size), "need": int(need),
            "all_above_picked": bool(ok_above), "chosen_tied": int(chosen_tied.size),
            "rule": classify(ranks, tied.size, need), "tied_ranks_head": ranks[:12].tolist(),
            "tied_ranks_tail": ranks[-4:].tolist()}


def canonical(values: np.ndarray, k: int) -> np.ndarray:
    order = np.lexsort((np.arange(values.size), -values.astype(np.float64)))
    return np.sort(order[:k])


def rtk_select(score32: np.ndarray, mask: np.ndarray, keep: int, chipwide: bool) -> np.ndarray:
    L = score32.size
    sc = torch.from_numpy(score32.copy()).to(DEV)
    mk = torch.from_numpy(mask).to(DEV)
    keep_idx = torch.empty(keep, dtype=torch.int64, device=DEV)
    rank = torch.empty(L, dtype=torch.int32, device=DEV)
    wsb = nv.lib.rtk_pivotkv_select_workspace_bytes(L)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV) if chipwide else None
    nv.check(nv.lib.rtk_pivotkv_select(nv.ptr(sc), nv.ptr(mk), L, keep, None, 0, 0, nv.ptr(keep_idx), nv.ptr(rank), None, keep,
                                       nv.ptr(ws), wsb if chipwide else 0, nv.stream()), "rtk_pivotkv_select")
    torch.cuda.synchronize()
    return keep_idx.cpu().numpy()


def pivotkv_cases():
    rng = np.random.default_rng(5)
    for L, keep in [(2304, 576), (6272, 1568), (6272, 624), (2304, 2303), (640, 1), (100352, 25088)]:
        # (a) fp32 scores, masked tokens forced to 1.0, the k-th value IS 1.0: n_above < keep < n_above + n_masked
        s = rng.normal(1.0, 0.2, size=L).astype(np.float32)
        mask = rng.uniform(size=L) < 1 / 3
        n_above = int(((s > 1.0) & ~mask).sum())
        if not (n_above < keep < n_above + mask.sum()):   # move the unmasked scores so that 1.0 straddles the boundary
            un = np.flatnonzero(~mask)
            order = un[np.argsort(-s[un])]
            want_above = max(0, min(keep - max(1, min(keep // 3, int(mask.sum()) // 2)), order.size))
            s[order[:want_above]] = 1.0 + np.abs(s[order[:want_above]] - 1.0) + 1e-3
            s[order[want_above:]] = 1.0 - np.abs(s[order[want_above:]] - 1.0) - 1e-3
        yield f"mask-ties fp32 L={L} keep={keep}", s, mask, keep, torch.float32
        # (b) bf16-valued scores: a run of equal values exactly at the threshold, no mask involvement
        for dt in (torch.bfloat16, torch.float16, torch.float32):
            sb = torch.from_numpy(rng.normal(1.0, 0.03, size=L).astype(np.float32)).to(torch.bfloat16).float().numpy()
            yield f"bf16-valued ties {str(dt)[6:]} L={L} keep={keep}", sb, np.zeros(L, bool), keep, dt
        # (c) both: mask ties and value ties, threshold wherever it falls
        sc = torch.from_numpy(rng.normal(1.0, 0.03, size=L).astype(np.float32)).to(torch.bfloat16).float().numpy()
        yield f"bf16-valued + mask bf16 L={L} keep={keep}", sc, mask, keep, torch.bfloat16


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "aten_ties.txt")
    lines = []

    def say(s=""):
        print(s)
        lines.append(s)

    say(f"# ATen topk tie order on {torch.cuda.get_device_name(0)}, torch {torch.__version__}, hip {torch.version.hip}")
    say("# reference expressions: longvideo_cache.py:272-277 (PivotKV), visual_compression.py:134-135 / :167-168 (DPSelect)")
    say()
    say("## PivotKV: s.masked_fill_(mask, 1.); s.topk(keep).indices.sort().values")
    summary = {"pivotkv": [], "dpselect": []}
    for name, s32, mask, keep, dt in pivotkv_cases():
        s = torch.from_numpy(s32).to(DEV).to(dt)
        m = torch.from_numpy(mask).to(DEV)
        s.masked_fill_(m, 1.)
        picked = s.topk(keep).indices.sort().values.cpu().numpy()
        seen = s.float().cpu().numpy()
        rep = tie_report(seen, picked, keep)
        canon = canonical(seen, keep)
        same_canon = bool(np.array_equal(picked, canon))
        # the library's kernels see fp32 scores (what its scoring passes produce) + the mask
        s_in = s32.copy() if dt is torch.float32 else s.float().cpu().numpy()
        if dt is not torch.float32:   # the override already happened in the 16-bit tensor; values are exact in fp32
            pass
        rtk_one = rtk_select(s_in, mask, keep, False)
        rtk_chip = rtk_select(s_in, mask, keep, True)
        rep.update(name=name, equals_lowest_index_first=same_canon, rtk_one_workgroup_equals_torch=bool(np.array_equal(rtk_one, picked)),
                   rtk_chipwide_equals_torch=bool(np.array_equal(rtk_chip, picked)))
        summary["pivotkv"].append(rep)
        say(f"{name}: kth={rep['kth']:.6f} above={rep['n_above']} tied={rep['n_tied']} need={rep['need']} -> {rep['rule']}"
            f" (== lowest-index-first set: {same_canon}); rtk_pivotkv_select == torch: one-workgroup {rep['rtk_one_workgroup_equals_torch']},"
            f" chip-wide {rep['rtk_chipwide_equals_torch']}; tied ranks chosen {rep['tied_ranks_head']} .. {rep['tied_ranks_tail']}")
    say()
    say("## DPSelect: keys.topk(k, sorted=False, dim=-1) then sort  (keys = dis with +2 on peaks; rows of [N, T] / one [T] row)")
    rng = np.random.default_rng(9)
    for T, N, tgt, sync in [(2048, 196, 512, False), (2048, 196, 1024, False), (2048, 196, 512, True), (256, 144, 128, False),
                            (64, 16, 16, False), (2048, 729, 204, False), (20, 7, 5, False), (20, 7, 5, True)]:
        # plateau distances: few distinct values -> ties among peaks, among non-peaks and across the boundary
        dis = (rng.integers(0, 6, size=(T, N)).astype(np.float32) / 8.0)
        dis[0] = 1.0
        dt_ = torch.from_numpy(dis).to(DEV)
        idx = torch.empty((tgt,) if sync else (tgt, N), dtype=torch.int64, device=DEV)
        mk = torch.empty((tgt, N), dtype=torch.bool, device=DEV)
        keys = torch.empty((2, T) if sync else (N, T), dtype=torch.float32, device=DEV)
        nv.check(nv.lib.rtk_dpselect_select(nv.ptr(dt_), T, N, tgt, 3, int(sync), nv.ptr(idx), nv.ptr(mk), nv.ptr(keys),
                                            nv.stream()), "rtk_dpselect_select")
        torch.cuda.synchronize()
        kk = keys[0] if sync else keys            # [T] / [N, T]: what the reference hands to topk
        _, ti = kk.topk(tgt, sorted=False, dim=-1)
        ti = ti.sort(dim=-1).values
        t_idx = ti if sync else ti.transpose(0, 1)
        same = bool(torch.equal(t_idx, idx))
        rows = kk.unsqueeze(0) if sync else kk
        rules = {}
        kn, pn = rows.cpu().numpy(), (ti.unsqueeze(0) if sync else ti).cpu().numpy()
        for r in range(rows.shape[0]):
            rule = tie_report(kn[r], pn[r], tgt)["rule"]
            rules[rule] = rules.get(rule, 0) + 1
        n_diff = int((t_idx != idx).any(dim=0).sum().item()) if not sync else int(not same)
        summary["dpselect"].append({"T": T, "N": N, "tgt": tgt, "sync": sync, "rtk_equals_torch": same, "rows_by_rule": rules,
                                    "rows_differing": n_diff})
        say(f"T={T} N={N} tgt={tgt} sync={sync}: rows by torch's tie rule {rules}; rtk_dpselect_select == torch: {same}"
            f" ({n_diff} rows differ)")
    say()
    say("JSON " + json.dumps(summary))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
