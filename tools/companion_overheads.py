"""Non-pass time of bench.py companions: for every bench JSON given, the headline value and, per companion block,
(frames/s, ms per step outside the two score passes, per-step wall times).  A companion whose non-pass time is far above
~70 ms (BASELINE geometry) / ~40 ms (real geometry) carried host or allocator stalls.

    python tools/companion_overheads.py gpurun_out/b1.json [...]
"""
import json,sys
for f in sys.argv[1:]:
    for line in open(f):
        line=line.strip()
        if line.startswith('{'):
            j=json.loads(line)
            row=[f.split('/')[-1], round(j['value'])]
            for k in ('fast_rounding','fp16_dtype','reference_rounding','real_geometry'):
                if k in j:
                    c=j[k]; tot=sum(v['total_ms'] for v in c['kernels_timed_region'].values())/c['steps']
                    row.append((k[:4], round(c["value"]), round(c["ms_per_step"]-tot,1), [round(x) for x in c.get("step_ms", [])]))
            print(row)
