"""One part of BASELINE config 3 (tools/ragged_parts.py) streamed on its own, for rocprofv3 passes:
    python tools/ragged_part_one.py {two-node|faced|massive|faced32|massive32} [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from heat_amd import HeatBatch, modeldict as mdl
which = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
md, st = mdl.ragged_mixed(S, Z=max(1, S // 100), dt=45.0, seed=20260401)
off = np.asarray(md["node_offset"]); n = np.diff(off)
m0 = np.asarray(md["mass"])[off[:-1]]
idx = {"two-node": n == 2, "faced": (m0 < 1e-5) & (n > 2), "massive": m0 >= 1e-5, "faced32": (m0 < 1e-5) & (n > 2) & (n <= 32),
       "massive32": (m0 >= 1e-5) & (n <= 32)}[which]
sub = mdl.subset(md, np.nonzero(idx)[0])
w = mdl.weather_series(10, 45.0)
with HeatBatch(sub, no_fusion=True) as b:
    b.upload_state(st)
    b.march_resident(w); b.synchronize()
    b.set_timing(True)
    b.march_resident(w); b.synchronize()
    print(which, b.class_counts(), b.get_timing())
