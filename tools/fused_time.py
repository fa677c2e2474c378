import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = int(sys.argv[1]); n = int(sys.argv[2]); P = int(sys.argv[3])
md, st = mdl.uniform_massive(S, n, Z=max(1, S // 100), dt=45.0)
w = mdl.weather_series(P, 45.0)
with HeatBatch(md, nodes_per_lane=int(sys.argv[4]) if len(sys.argv) > 4 else 0) as b:
    b.upload_state(st)
    b.march_resident(w)
    try:
        b.synchronize()
    except Exception as e:
        pass
    b.set_timing(True)
    for _ in range(5):
        b.march_resident(w)
    try:
        b.synchronize()
    except Exception:
        pass
    su, ss, ns = b.get_timing()
    print(f"fused surfaces {b.n_fused_surfaces}: {su:.1f} us/substep")
