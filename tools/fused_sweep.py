"""Cluster-resident march: time per sub-timestep against the sub-timesteps per march call (1 M x 32)."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
md, st = mdl.uniform_massive(S, n, Z=max(1, S // 100), dt=45.0)
with HeatBatch(md, use_graph=True) as b:
    b.upload_state(st)
    for P in (1, 2, 5, 10, 20, 50, 100):
        w = mdl.weather_series(P, 45.0)
        calls = max(2, 200 // P)
        b.march_resident(w); b.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            b.march_resident(w)
        b.synchronize()
        el = time.perf_counter() - t0
        print(f"P={P:4d}: {el / (calls * P) * 1e6:7.1f} us per sub-timestep -> {S * n * calls * P / el:.3e} node-updates/s", flush=True)
