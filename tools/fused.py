"""Cluster-resident march vs streamed march on the headline workload: equality of results and time per sub-timestep."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from heat_amd import HeatBatch, modeldict as mdl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
P = int(sys.argv[3]) if len(sys.argv) > 3 else 20
md, st = mdl.uniform_massive(S, n, Z=max(1, S // 100), dt=45.0)
w = mdl.weather_series(P, 45.0)
res = {}
for npl in (0, 8, 16):
    with HeatBatch(md, nodes_per_lane=npl) as b:
        print("npl", npl, "classes", b.class_counts(), "fused surfaces", b.n_fused_surfaces, flush=True)
        for fused in (False, True):
            b.set_fusion(fused)
            b.upload_state(st)
            b.march_resident(w); b.synchronize()
            out = st.copy(); b.download_state(out)
            res[(npl, fused)] = out
            b.set_timing(True)
            t0 = time.perf_counter()
            for _ in range(5):
                b.march_resident(w)
            b.synchronize()
            el = time.perf_counter() - t0
            su, ss, ns = b.get_timing()
            b.set_timing(False)
            print(f"  fused={fused}: surf {su:.1f} us/substep, substep {ss:.1f} us, wall {el/(5*P)*1e6:.1f} us/substep -> {S*n*5*P/el:.3e} node-updates/s", flush=True)
        d = np.abs(res[(npl, True)] - res[(npl, False)]).max()
        print("  max |fused - streamed| =", d, flush=True)
