"""Clusters larger than a workgroup (buildings of many rooms joined by partitions): teams of workgroups vs everything
streamed, 1 M walls:   python tools/teams.py [ROOMS N] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = 1_000_000
args = [int(x) for x in sys.argv[1:]] or [40, 32, 24, 32, 40, 20, 16, 32]
w = mdl.weather_series(20, 45.0)
for rooms, n in zip(args[0::2], args[1::2]):
    md, st = mdl.partitioned_buildings(S, n, rooms=rooms, dt=45.0, seed=5)
    out = []
    for kw in (dict(no_fusion=True), dict()):
        with HeatBatch(md, use_graph=True, **kw) as b:
            b.upload_state(st)
            b.march_resident(w); b.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                b.march_resident(w)
            b.synchronize()
            out.append(((time.perf_counter() - t0) / 80 * 1e6, b.class_counts(), b.n_fused_surfaces))
    print("rooms=%d n=%d (%d walls, %d zones): streamed %.1f us %s | planner %.1f us %s fused %d" % (
        rooms, n, md["n_surfaces"], md["n_zones"], out[0][0], out[0][1], out[1][0], out[1][1], out[1][2]), flush=True)
