"""The planner's cost model against measurement: planner's choice vs everything streamed, several workloads at 1 M surfaces."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = 1000000
def run(name, md, st):
    w = mdl.weather_series(20, 45.0)
    out = []
    for kw in (dict(no_fusion=True), dict()):
        with HeatBatch(md, use_graph=True, **kw) as b:
            b.upload_state(st)
            b.march_resident(w); b.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                b.march_resident(w)
            b.synchronize()
            us = (time.perf_counter() - t0) / 80 * 1e6
            out.append((us, b.class_counts(), b.n_fused_surfaces))
    print(f"{name}: streamed {out[0][0]:.1f} us {out[0][1]} | planner {out[1][0]:.1f} us {out[1][1]} fused {out[1][2]}", flush=True)
for n in (32, 20, 13, 10, 8, 6, 48, 64):
    md, st = mdl.uniform_massive(S, n, Z=S // 100, dt=45.0)
    run(f"uniform n={n}", md, st)
for name, gen in (("clustered", mdl.clustered_massive), ("rooms", mdl.rooms_with_windows), ("ragged", mdl.ragged_mixed)):
    md, st = gen(S, dt=45.0)
    run(name, md, st)
