"""A/B of two builds of the library on the planner's choice, uniform walls (HEAT_AMD_LIB selects the build):
    python tools/ab.py N [N ...]          prints us per sub-timestep at 20 per call, 1 M walls of N nodes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = 1000000
for n in [int(x) for x in sys.argv[1:]] or [32]:
    md, st = mdl.uniform_massive(S, n, Z=S // 100, dt=45.0)
    w = mdl.weather_series(20, 45.0)
    with HeatBatch(md, use_graph=True) as b:
        b.upload_state(st)
        b.march_resident(w); b.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(4):
                b.march_resident(w)
            b.synchronize()
            ts.append((time.perf_counter() - t0) / 80 * 1e6)
        print("%s n=%d: %s us per sub-timestep, classes %s fused %d" % (
            os.path.basename(os.environ.get("HEAT_AMD_LIB", "libheat_amd.so")), n, " / ".join("%.1f" % t for t in ts),
            b.class_counts(), b.n_fused_surfaces), flush=True)
