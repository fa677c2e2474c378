"""Streamed vs planner's choice vs everything-that-can-be fused, 1 M surfaces: python tools/fuse_compare.py [workloads...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = 1000000
gens = {"clustered": lambda: mdl.clustered_massive(S, dt=45.0), "rooms": lambda: mdl.rooms_with_windows(S, dt=45.0),
        "partitions": lambda: mdl.partitioned_buildings(S, 32, dt=45.0), "partitions20": lambda: mdl.partitioned_buildings(S, 20, dt=45.0),
        "uniform32": lambda: mdl.uniform_massive(S, 32, Z=S // 100, dt=45.0)}
for name in (sys.argv[1:] or ["clustered", "partitions"]):
    md, st = gens[name]()
    w = mdl.weather_series(20, 45.0)
    res = []
    for label, kw in (("streamed", dict(no_fusion=True)), ("planner", dict()), ("fuse all", dict(fuse_always=True))):
        with HeatBatch(md, use_graph=True, **kw) as b:
            b.upload_state(st)
            b.march_resident(w); b.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                b.march_resident(w)
            b.synchronize()
            res.append("%s %.1f us (fused %d, classes %s)" % (label, (time.perf_counter() - t0) / 100 * 1e6, b.n_fused_surfaces, b.class_counts()))
    print(name + ": " + " | ".join(res), flush=True)
