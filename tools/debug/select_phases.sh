#!/bin/bash
# Where do the 29 us of pivotkv_select_units_kernel go?  Builds a throw-away copy of csrc/ whose select_fast_body stamps
# s_memrealtime (100 MHz) at its phase boundaries into the unit's workspace, runs 28 units and prints the deltas.
#   (here)  tools/debug/select_phases.sh build      (GPU box)  tools/debug/select_phases.sh run
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = build ]; then
  tmp=$ROOT/gpurun_out/_seltime_src; rm -rf $tmp; mkdir -p $tmp; cp $ROOT/video-retake_amd/csrc/*.* $ROOT/video-retake_amd/csrc/Makefile $tmp/
  python3 - $tmp/pivotkv_evict.hip <<'PY'
import re, sys
p = sys.argv[1]; s = open(p).read()
a = s.index("__device__ __forceinline__ void select_lds_body"); b = s.index("// ------------------------------------------------------------------------------------------------\n// Chip-wide selection")
body = s[a:b]
body = body.replace("int64_t* __restrict__ pos_out, int64_t pos_ld) {", "int64_t* __restrict__ pos_out, int64_t pos_ld, unsigned long long* stamp) {\n    int sn = 0;\n#define STAMP() do { if (threadIdx.x == 0 && stamp) stamp[sn++] = wall_clock64(); } while (0)\n    STAMP();", 1)
body = body.replace("    // ---- 1: exact k-th largest key", "    STAMP();\n    // ---- 1: exact k-th largest key", 1)
body = body.replace("        pmask |= 255u << shift;\n    }", "        pmask |= 255u << shift;\n        STAMP();\n    }", 1)
body = body.replace("    // ---- 3: kept indices, ascending", "    STAMP();\n    // ---- 3: kept indices, ascending", 1)
body = body.replace("    // ---- 4: min_temp_id", "    STAMP();\n    // ---- 4: min_temp_id", 1)
body = body.replace("    // ---- 5: outputs", "    STAMP();\n    // ---- 5: outputs", 1)
body = body[:body.rindex("}")] + "    __syncthreads();\n    STAMP();\n}\n\n"
s = s[:a] + body + s[b:]
s = s.replace("select_lds_body(u.score, u.mask, L, keep, u.pos, P, reforge, u.keep_idx, u.rank, u.pos_out, pos_ld);", "select_lds_body(u.score, u.mask, L, keep, u.pos, P, reforge, u.keep_idx, u.rank, u.pos_out, pos_ld, (unsigned long long*)u.workspace);")
open(p, "w").write(s)
PY
  RTK_SRC=$tmp $ROOT/tools/variants.sh seltime ""
  exit 0
fi
export RETAKE_HIP_LIB=$ROOT/video-retake_amd/retake/_lib/variants/libretake_hip_seltime.so
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "video-retake_amd"))
import torch
import retake._native as nv
dev = torch.device("cuda:0")
for L, keep in ((6272, 1568), (2304, 576)):
    n, P = 28, 3
    units = (nv.SelectUnit * n)(); hold = []
    pos_new = torch.zeros((P, n, keep), dtype=torch.int64, device=dev)
    for i in range(n):
        score = torch.rand(L, device=dev) * 0.2 + 0.9
        mask = torch.rand(L, device=dev) < 0.33
        pos = torch.stack([torch.arange(L, device=dev) // 196 + 40, torch.arange(L, device=dev) % 14, torch.arange(L, device=dev) % 14]).contiguous()
        ki = torch.empty(keep, dtype=torch.int64, device=dev); ws = torch.zeros(max(4096, L * 2), dtype=torch.uint8, device=dev)
        u = units[i]
        u.partial, u.score, u.mask, u.pos = None, score.data_ptr(), mask.data_ptr(), pos.data_ptr()
        u.keep_idx, u.rank, u.pos_out, u.workspace = ki.data_ptr(), None, pos_new.data_ptr() + i * keep * 8, ws.data_ptr()
        hold.append((score, mask, pos, ki, ws))
    for it in range(3):
        nv.check(nv.lib.rtk_pivotkv_select_batched(units, n, 4, 1, 7, L, keep, P, 1, n * keep, 0, nv.stream()), "sel")
    torch.cuda.synchronize()
    names = ["keys", "radix 24", "radix 16", "radix 8", "radix 0", "scan", "keepL", "min", "outputs"]
    for i in (0, 13, 27):
        st = hold[i][4][:80].view(torch.int64).tolist()
        d = [(st[j + 1] - st[j]) * 0.01 for j in range(9)]
        print("L=%d unit %2d: " % (L, i) + "  ".join("%s %.2f" % (nm, x) for nm, x in zip(names, d)) + "  | total %.2f us" % ((st[9] - st[0]) * 0.01))
PY
