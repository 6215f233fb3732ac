import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/video-retake_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import retake._native as nv
from oracle import oracle as orc
import synth
dev = torch.device("cuda:0")
# bit-level check of the bf16 distance against the oracle's true-division chain on video-like and iid data
for kind, T, N, C in (("video", 96, 32, 1280), ("iid", 64, 40, 1152), ("video", 40, 7, 3584)):
    x = synth.make_frames(kind, 5, T, N, C)[0]
    xb = torch.from_numpy(x).bfloat16()
    xo = xb.view(torch.int16).numpy().view(np.uint16)
    d_or = orc.dpselect_dis(xo)
    xt = xb.to(dev)
    dis = torch.empty((T, N), dtype=torch.float32, device=dev)
    nv.check(nv.lib.rtk_dpselect_dis(nv.ptr(xt), T, N, C, nv.RTK_BF16, nv.ptr(dis), nv.stream()), "dis")
    d = dis.cpu().numpy()
    diff = np.abs(d - d_or)
    print(kind, T, N, C, "max diff", diff.max(), "frac differing", (diff > 0).mean())
# timing at the BASELINE size
x = torch.randn((2048, 196, 1280), device=dev).bfloat16()
dis = torch.empty((2048, 196), dtype=torch.float32, device=dev)
for _ in range(3):
    nv.check(nv.lib.rtk_dpselect_dis(nv.ptr(x), 2048, 196, 1280, nv.RTK_BF16, nv.ptr(dis), nv.stream()), "dis")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    nv.check(nv.lib.rtk_dpselect_dis(nv.ptr(x), 2048, 196, 1280, nv.RTK_BF16, nv.ptr(dis), nv.stream()), "dis")
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print("dis bf16 2048x196x1280: %.1f us  %.2f TB/s" % (us, 2048*196*1280*2/us/1e6))
