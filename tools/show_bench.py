#!/usr/bin/env python3
"""Pretty-print a bench.py REPORT (bench_report.json; bench.py's stdout is only the short contract line):
`show_bench.py [FILE]` (default: bench_report.json at the repo root), or a report / line from stdin with `-`."""
import json
import os
import sys

if len(sys.argv) > 1 and sys.argv[1] != "-":
    text = open(sys.argv[1]).read()
elif len(sys.argv) > 1:
    text = sys.stdin.read()
else:
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_report.json")).read()
d = json.loads(text.strip().splitlines()[-1])
if "report" in d and "kernels_timed_region" not in d and os.path.exists(d["report"]):   # handed the short line: follow it
    d = json.loads(open(d["report"]).read())
print(f"value={d['value']:.1f} {d['unit']}  ms_per_step={d['ms_per_step']:.1f}  n_gpus={d['n_gpus']} dtype={d['dtype']}")
tot = 0.0
for k, v in (d.get("kernels") or d.get("kernels_untimed_single_stream", {})).items():
    print(f"  {k:18s} n={v['launches']:5d} avg={v['avg_us']:8.1f} us total={v['total_ms']:8.1f} ms")
    tot += v["total_ms"]
print(f"  kernel total {tot:.1f} ms (untimed single-stream segment)")
for k, v in d.get("kernels_timed_region", {}).items():
    print(f"  [timed] {k:18s} n={v['launches']:5d} avg={v['avg_us']:8.1f} us total={v['total_ms']:8.1f} ms")
for k in ("roofline", "roofline_hbm_kernels", "cpu_baseline", "speedup_vs_cpu_baseline"):
    if k in d:
        print(k, d[k])
comp = {k: d[k] for k in ("real_geometry", "no_keypatch_mask", "rotary_module_called", "reference_rounding", "fast_rounding",
                         "fp16_dtype", "fp32_parity_dtype") if k in d}
for g, c in (d.get("pre_rope_prologue") or {}).items():
    if isinstance(c, dict):
        comp["pre_rope_prologue/" + g] = c
for k, c in comp.items():
    if True:
        share = c.get("per_update_kernels_share_of_gpu_time")
        print(f"{k}: {c['value']:.1f} frames/s  ms_per_step={c['ms_per_step']:.1f}  roofline {c['roofline']['kernel']} "
              f"frac={c['roofline']['frac']:.3f}  " + " ".join(f"{n}={v['avg_us']:.0f}us" for n, v in c["kernels_timed_region"].items())
              + (f"  per-update kernels {100 * share:.1f} % of GPU time" if share is not None else ""))
if "llava_workload" in d:
    c = d["llava_workload"]
    print(f"llava_workload: {c['value']:.1f} frames/s  ms_per_step={c['ms_per_step']:.1f}  pass 1 of 2.5 PF {c['score_pass1_frac_of_2.5PF']:.3f}  "
          + " ".join(f"{n}={v['avg_us']:.0f}us" for n, v in c["kernels_timed_region"].items()))
if "memory" in d:
    m = d["memory"]
    print(f"memory: product peak {m['product_peak_bytes'] / 1e9:.2f} GB (allocator peak {m['peak_allocated_bytes'] / 1e9:.2f} - resident inputs "
          f"{m['resident_inputs_bytes'] / 1e9:.2f}); reference by formula {m['reference_peak_by_formula']['bytes'] / 1e9:.2f} GB "
          f"(x{m['product_peak_over_reference_formula']:.2f}); scratch / cache rows {m['scratch_over_cache_rows']:.2f}")
    print("  split GB:", {k: round(v / 1e9, 3) for k, v in m["split_bytes"].items()})
if "hbm_achievable" in d:
    print("hbm_achievable", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["hbm_achievable"].items() if k != "note"})
if "decode_prologue" in d:
    for k in ("decode_token", "text_segment_64"):
        c = d["decode_prologue"][k]
        print(f"decode_prologue/{k}: fused {c['fused_us_per_step']:.0f} us  op-by-op {c['op_by_op_us_per_step']:.0f} us  ({c['speedup']:.1f}x), "
              f"{d['decode_prologue']['layers']} layers")
for k in ("sharded_equals_sequential", "rccl_world_size", "p2p_world_size", "phase_ms", "phase_bytes_rank0"):
    if k in d:
        print(k, d[k])
if isinstance(d.get("n1_same_arithmetic"), dict):
    c = d["n1_same_arithmetic"]
    print("n1_same_arithmetic (sharded path at world size 1):", c.get("error") or
          f"{c['value']:.1f} frames/s  ms_per_step={c['ms_per_step']:.1f}  phase_ms {c['phase_ms']}")
