"""heat_batch_march on a caller-owned host state (the drop-in call): time per call by what comes back.
    python tools/pcie.py [S] [n] [P]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from heat_amd import HeatBatch, modeldict as mdl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
md, st = mdl.uniform_massive(S, n, Z=max(1, S // 100), dt=45.0)
w = mdl.weather_series(P, 45.0)
with HeatBatch(md, use_graph=True) as b:
    b.upload_state(st)
    t = time.perf_counter(); b.march_resident(w); b.synchronize(); t_res = time.perf_counter() - t
    t = time.perf_counter(); b.march_resident(w); b.synchronize(); t_res = time.perf_counter() - t
    for name, what in (("all outputs", b.OUT_ALL), ("scalars + zones (no node temperatures)", b.OUT_SCALARS | b.OUT_ZONES),
                       ("zones only", b.OUT_ZONES)):
        b.march(st, w, outputs=what)
        t = time.perf_counter()
        for _ in range(5):
            b.march(st, w, outputs=what)
        dt_ = (time.perf_counter() - t) / 5
        print(f"heat_batch_march_ex n_sub={P}, {name}: {dt_*1e3:.2f} ms per call (resident march alone {t_res*1e3:.2f} ms); state {st.nbytes/1e6:.0f} MB", flush=True)
    t = time.perf_counter(); b.upload_inputs(st); b.synchronize(); print(f"upload_inputs: {(time.perf_counter()-t)*1e3:.2f} ms")
    t = time.perf_counter(); b.download_outputs(st, b.OUT_ALL); print(f"download_outputs(all): {(time.perf_counter()-t)*1e3:.2f} ms")
    t = time.perf_counter(); b.download_outputs(st, b.OUT_SCALARS | b.OUT_ZONES); print(f"download_outputs(scalars+zones): {(time.perf_counter()-t)*1e3:.2f} ms")
