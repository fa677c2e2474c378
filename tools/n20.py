import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
md, st = mdl.uniform_massive(1000000, 20, dt=45.0)
w = mdl.weather_series(40, 45.0)
for npl in (0, 4, 8):
    with HeatBatch(md, nodes_per_lane=npl) as b:
        b.upload_state(st); b.march_resident(w[:10]); b.synchronize()
        b.set_timing(True); b.march_resident(w); b.synchronize()
        print("n=20 npl", npl, b.class_counts(), b.get_timing())
