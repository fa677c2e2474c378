"""Short march calls (ThermalModel::march with 1-6 sub-timesteps per call, the reference's validation models run 2 or 4):
time per call and per sub-timestep, streamed and by the planner's choice, 1 M x 32 headline; and the fixed cost of a call
(a model of 100 walls, where the kernels are a few microseconds).   python tools/short_calls.py [S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from heat_amd import HeatBatch, modeldict as mdl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000


def per_call(b, w, reps):
    b.march_resident(w); b.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.march_resident(w)
    b.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for s in (100, S):
    md, st = mdl.uniform_massive(s, n=32, Z=max(1, s // 100), dt=45.0)
    for label, kw in (("streamed", dict(no_fusion=True)), ("planner", dict())):
        with HeatBatch(md, use_graph=True, **kw) as b:
            b.upload_state(st)
            out = []
            for n_sub in (0, 1, 2, 3, 4, 5, 6, 10, 20):
                w = mdl.weather_series(n_sub, 45.0)
                us = per_call(b, w, 200 if s == 100 else 30)
                out.append("%d: %.0f us%s" % (n_sub, us, " (%.1f per sub-timestep)" % (us / n_sub) if n_sub else ""))
            print("%8d walls, %-8s | " % (s, label) + " | ".join(out), flush=True)
