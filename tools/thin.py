"""Thin walls (n <= 16) at scale: streamed vs fused with 8 or 4 nodes per lane."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = 1000000
for n in (13, 16, 10):
    md, st = mdl.uniform_massive(S, n, Z=S // 100, dt=45.0)
    w = mdl.weather_series(20, 45.0)
    for kw in (dict(no_fusion=True), dict(fuse_always=True, nodes_per_lane=8), dict(fuse_always=True, nodes_per_lane=4)):
        with HeatBatch(md, use_graph=True, **kw) as b:
            b.upload_state(st)
            b.march_resident(w); b.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                b.march_resident(w)
            b.synchronize()
            us = (time.perf_counter() - t0) / 100 * 1e6
            print(f"n={n} {kw}: classes {b.class_counts()} fused {b.n_fused_surfaces}: {us:.1f} us per sub-timestep -> {S*n/us*1e6:.3e} node-updates/s", flush=True)
