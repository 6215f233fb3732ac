import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "video-retake_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import golden_util as gu
import glue_stubs as gs
import retake.longvideo_cache as lc
import retake.qwen2_vl as q
dev = torch.device("cuda:0")
g = gu.load("glue_attention_qwen2vl")
S = float(g["attention_scaling"])
layers = [gs.StubAttention(l, 64, 4, 2, (2, 3, 3), S, weights=[g[f"w{l}_{i}"] for i in range(7)]).to_device(dev).eval() for l in range(2)]
llm = types.SimpleNamespace(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2)
llm.longvideo_kwargs = {"kvcache_compression": True, "kvcache_compression_kwargs": {"compression_ratio": 0.5, "compression_method": "pivotkv", "pos_embed_reforge": True}}
cache = lc.build_kvcache(llm)
total = 0
for si in range(int(g["n_steps"])):
    kind = str(g[f"s{si}_kind"])
    x = torch.from_numpy(g[f"s{si}_x"]).to(dev)
    n = x.shape[1]; total += n
    mask4 = torch.from_numpy(g[f"s{si}_mask4"]).to(dev)
    cp = torch.arange(total - n, total, device=dev)
    cache.kvcache_compression = kind == "video"
    cache.keypatches_mask_chunk = torch.from_numpy(g[f"s{si}_kpmask"]).to(dev) if kind == "video" else None
    pos_shared = torch.from_numpy(g[f"s{si}_pos_in"]).to(dev)
    print("step", si, kind, "pos_in", g[f"s{si}_pos_in"][0, 0, :3], "n", n)
    for l, att in enumerate(layers):
        with torch.no_grad():
            o = q.retake_Qwen2VLAttention_forward(att, x, mask4, pos_shared, cache, False, True, cp)
        ref = g[f"s{si}_l{l}_out"]
        err = np.abs(o[0].cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        st = cache._layers[l]
        print("  layer", l, "err", err, "got", pos_shared[0, 0, :2].tolist(), "want", g[f"s{si}_l{l}_pos_after"][0, 0, :2].tolist(),
              "pos_len", st.pos_len, "pending", st.pending, "pos_layers", cache._pos_layers,
              "pos tail", st.pos[0, max(0, st.pos_len - 3):st.pos_len].tolist() if st.pos is not None else None)
    cache.after_forward()
    for l in range(2):
        st = cache._layers[l]
        print("  after flush layer", l, "pos_len", st.pos_len, "tail", st.pos[0, max(0, st.pos_len - 3):st.pos_len].tolist())
