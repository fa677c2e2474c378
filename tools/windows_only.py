"""100 000 double-glazed windows (4 no-mass nodes around a gas cavity) on their own, streamed: the workload of
k_surfaces_small<1> for counter passes.  python tools/windows_only.py [S] [P]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10
md, st = mdl.glazing_cavity(S, Z=max(1, S // 100), dt=45.0, trombe_fraction=0.0)
w = mdl.weather_series(P, 45.0)
with HeatBatch(md, use_graph=True, no_fusion=True) as b:
    b.upload_state(st)
    b.march_resident(w); b.synchronize()
    i0 = b.nomass_iterations()
    b.set_timing(True)
    b.march_resident(w); b.synchronize()
    su, ss, n = b.get_timing()
    print("%d windows: surfaces %.1f us, sub-timestep %.1f us, %.2f passes per window and sub-timestep" % (
        S, su, ss, (b.nomass_iterations() - i0) / P / S))
