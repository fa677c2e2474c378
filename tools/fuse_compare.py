"""Streamed vs planner's choice vs everything-that-can-be fused, 1 M surfaces: python tools/fuse_compare.py [workloads...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = 1000000
def campus(share):
    """uniform 32-node walls, zones of 100; the first `share` of the zones chained by one wall in ten into a single
    cluster (a campus joined by interior walls: far too large for a workgroup), the rest isolated houses"""
    import numpy as np
    md, st = mdl.uniform_massive(S, 32, Z=S // 100, dt=45.0)
    Z = S // 100
    zc = int(Z * share)
    chain = (md["back_zone"] < zc) & (np.arange(S) % 10 == 0)
    md["front_kind"] = np.where(chain, mdl.SPACE, md["front_kind"]).astype(np.int32)
    md["front_zone"] = np.where(chain, (md["back_zone"] + 1) % max(zc, 1), md["front_zone"]).astype(np.int32)
    return md, st


gens = {"campus30": lambda: campus(0.3), "campus70": lambda: campus(0.7),"clustered": lambda: mdl.clustered_massive(S, dt=45.0), "rooms": lambda: mdl.rooms_with_windows(S, dt=45.0),
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
