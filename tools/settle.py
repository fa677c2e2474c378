"""How long the chip takes to settle at its clock under this load: us per sub-timestep of consecutive blocks of march calls
from a cold start, and again after two seconds of idling.   python tools/settle.py [CONFIG] [plan|stream]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from heat_amd import HeatBatch, modeldict as mdl
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
mode = sys.argv[2] if len(sys.argv) > 2 else "plan"
md, st, _ = bench.build_config(cfg, types.SimpleNamespace(surfaces=1_000_000, nodes=32), 45.0, 20260401)
w = mdl.weather_series(20, float(md["dt"]))
with HeatBatch(md, use_graph=True, no_fusion=(mode == "stream")) as b:
    b.upload_state(st)
    b.march_resident(w); b.synchronize()
    t_start = time.perf_counter()
    for block in range(10):
        t0 = time.perf_counter()
        for _ in range(5):
            b.march_resident(w)
        b.synchronize()
        t1 = time.perf_counter()
        print("%s [%s] %6.1f ms after the start: %.2f us per sub-timestep" % (cfg, mode, (t0 - t_start) * 1e3, (t1 - t0) / 100 * 1e6), flush=True)
    time.sleep(2.0)
    t0 = time.perf_counter()
    for _ in range(5):
        b.march_resident(w)
    b.synchronize()
    print("after 2 s idle: %.2f us per sub-timestep" % ((time.perf_counter() - t0) / 100 * 1e6))
