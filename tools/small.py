"""Small batches (launch-bound when streamed): time per sub-timestep, auto plan vs no fusion."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
for S, n, Z in ((1, 13, 1), (100, 13, 4), (2000, 13, 40), (2000, 28, 40), (8000, 20, 80)):
    md, st = mdl.uniform_massive(S, n, Z=Z, dt=90.0)
    w = mdl.weather_series(20, 90.0)
    for nofuse in (True, False):
        with HeatBatch(md, no_fusion=nofuse, use_graph=True) as b:
            b.upload_state(st)
            b.march_resident(w); b.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                b.march_resident(w)
            b.synchronize()
            us = (time.perf_counter() - t0) / (20 * len(w)) * 1e6
            print(f"S={S} n={n}: {'streamed' if nofuse else 'auto    '} classes {b.class_counts()} fused {b.n_fused_surfaces}: {us:.2f} us per sub-timestep", flush=True)
