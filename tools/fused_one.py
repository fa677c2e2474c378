"""One workload through the cluster-resident march (for rocprofv3 passes): S surfaces x n nodes, P sub-timesteps per march."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
S = int(sys.argv[1]); n = int(sys.argv[2]); P = int(sys.argv[3]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
fused = (sys.argv[5] != "stream") if len(sys.argv) > 5 else True
md, st = mdl.uniform_massive(S, n, Z=max(1, S // 100), dt=45.0)
w = mdl.weather_series(P, 45.0)
with HeatBatch(md) as b:
    b.set_fusion(fused)
    b.upload_state(st)
    for _ in range(reps):
        b.march_resident(w)
    b.synchronize()
    print("classes", b.class_counts(), "fused", b.n_fused_surfaces)
