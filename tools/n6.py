import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = 1000000
md, st = mdl.uniform_massive(S, 6, Z=S // 100, dt=45.0)
w = mdl.weather_series(20, 45.0)
for rep in range(2):
    for kw in (dict(no_fusion=True), dict()):
        with HeatBatch(md, use_graph=True, **kw) as b:
            b.upload_state(st)
            b.march_resident(w); b.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                b.march_resident(w)
            b.synchronize()
            us = (time.perf_counter() - t0) / 80 * 1e6
            print(kw, b.class_counts(), b.n_fused_surfaces, "%.1f us" % us, flush=True)
