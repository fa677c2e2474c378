import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heat_amd import HeatBatch, modeldict as mdl
from oracle import oracle as orc

def probe(name, md, st, nsub, **opts):
    w = mdl.weather_series(nsub, md["dt"])
    ref = st.copy()
    t = time.time(); rc, it = orc.OracleModel(md).march(ref, w); tc = time.time() - t
    got = st.copy()
    with HeatBatch(md, **opts) as b:
        b.upload_state(got)
        t = time.time(); b.march_resident(w); b.synchronize(); tg = time.time() - t
        b.set_timing(True)
        b.march_resident(w); b.synchronize()
        surf_us, sub_us, n = b.get_timing()
        b.set_timing(False)
        b.upload_state(got)
        b.march(got, w)
        cc = b.class_counts(); ab = b.algorithmic_bytes; nn = b.n_nodes
    ns = mdl.node_slots(md)
    d = np.abs(got[ns] - ref[ns]); rel = d / (1e-9 + 1e-9 * np.abs(ref[ns]))
    print(f"{name}: classes {cc} nodes {nn} | max|dT| {d.max():.3e} (tol-units {rel.max():.3e}) | oracle {tc:.3f}s gpu-wall {tg*1e3:.2f} ms | surf kernels {surf_us:.1f} us/substep, substep {sub_us:.1f} us | {ab/surf_us/1e3:.1f} GB/s algorithmic | {nn/ (sub_us*1e-6)/1e9:.2f} G node-updates/s", flush=True)

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for npl in (4, 8, 16):
    md, st = mdl.uniform_massive(S, 32, dt=45.0)
    probe(f"uniform32 S={S} npl={npl}", md, st, 20, nodes_per_lane=npl)
md, st = mdl.uniform_massive(10000, 20, Z=100, dt=90.0, identical=True, vertical=True)
probe("config2", md, st, 50)
md, st = mdl.ragged_mixed(S, dt=45.0)
probe(f"ragged S={S}", md, st, 10)
md, st = mdl.glazing_cavity(20000, dt=45.0)
probe("glazing S=20000", md, st, 10)
