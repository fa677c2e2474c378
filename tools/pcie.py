import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from heat_amd import HeatBatch, modeldict as mdl
md, st = mdl.uniform_massive(1000000, 32, dt=45.0)
w = mdl.weather_series(100, 45.0)
with HeatBatch(md, use_graph=True) as b:
    b.upload_state(st)
    b.march(st, w[:2])
    for n in (1, 10, 100):
        t = time.perf_counter(); b.march(st, w[:n]); dt_ = time.perf_counter() - t
        print(f"heat_batch_march n_sub={n}: {dt_*1e3:.1f} ms wall -> {32e6*n/dt_/1e9:.2f} G node-updates/s (PCIe-inclusive); state {st.nbytes/1e6:.0f} MB")
    t = time.perf_counter(); b.march_resident(w); b.synchronize(); dt_ = time.perf_counter() - t
    print(f"resident n_sub=100: {dt_*1e3:.1f} ms -> {32e6*100/dt_/1e9:.2f} G node-updates/s")
