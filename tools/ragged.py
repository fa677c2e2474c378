import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from heat_amd import HeatBatch, modeldict as mdl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
which = sys.argv[2] if len(sys.argv) > 2 else "ragged"
if which == "ragged":
    md, st = mdl.ragged_mixed(S, dt=45.0)
elif which == "rooms":
    md, st = mdl.rooms_with_windows(S, dt=45.0)
elif which == "clustered":
    md, st = mdl.clustered_massive(S, dt=45.0)
elif which == "glazing":
    md, st = mdl.glazing_cavity(S, dt=45.0)
else:
    md, st = mdl.uniform_massive(S, 32, dt=45.0)
w = mdl.weather_series(20, 45.0)
nofuse = len(sys.argv) > 3 and sys.argv[3] == "nofuse"
always = len(sys.argv) > 3 and sys.argv[3] == "always"
with HeatBatch(md, no_fusion=nofuse, fuse_always=always, use_graph=True) as b:
    b.upload_state(st)
    b.march_resident(w); b.synchronize()
    b.set_timing(True); b.march_resident(w); b.synchronize()
    t = b.get_timing()
    import time
    b.set_timing(False)
    b.march_resident(w); b.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        b.march_resident(w)
    b.synchronize()
    wall = (time.perf_counter() - t0) / (5 * len(w)) * 1e6
    print(which, S, "no_fusion" if nofuse else ("always" if always else "auto"), b.class_counts(), "fused", b.n_fused_surfaces,
          "events (surf, substep, n):", tuple(round(x, 1) for x in t), "wall us/substep: %.1f" % wall)
