import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from heat_amd import HeatBatch, modeldict as mdl
S = 1000000
md, st = mdl.uniform_massive(S, 32, dt=45.0)
w = mdl.weather_series(60, 45.0)
for npl in (4, 8, 16):
    for nopal in (False, True):
        with HeatBatch(md, nodes_per_lane=npl, no_palette=nopal) as b:
            b.upload_state(st)
            b.march_resident(w[:10]); b.synchronize()
            b.set_timing(True); b.march_resident(w); b.synchronize()
            su, ss, n = b.get_timing()
            print(f"npl={npl} palette={not nopal}: surf {su:.1f} us  substep {ss:.1f} us  -> {b.algorithmic_bytes/su/1e3:.0f} GB/s algorithmic", flush=True)
