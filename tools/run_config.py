"""One BASELINE config through the streamed / planned march, for rocprofv3 passes and quick timings:
    python tools/run_config.py CONFIG [P] [REPS] [fusion: plan|stream] [S] [nodes per lane]"""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from heat_amd import HeatBatch, modeldict as mdl
cfg = sys.argv[1]
P = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mode = sys.argv[4] if len(sys.argv) > 4 else "plan"
S = int(sys.argv[5]) if len(sys.argv) > 5 else 1_000_000
npl = int(sys.argv[6]) if len(sys.argv) > 6 else 0
args = types.SimpleNamespace(surfaces=S, nodes=32)
md, st, desc = bench.build_config(cfg, args, 45.0, 20260401)
dt = float(md["dt"])
w = mdl.weather_series(P, dt)
with HeatBatch(md, use_graph=True, no_fusion=(mode == "stream"), nodes_per_lane=npl) as b:
    b.upload_state(st)
    b.march_resident(w); b.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.march_resident(w)
    b.synchronize()
    wall = (time.perf_counter() - t0) / (reps * P) * 1e6
    b.set_timing(True)
    b.march_resident(w); b.synchronize()
    su, ss, ns = b.get_timing()
    ab = b.algorithmic_bytes
    print("%s [%s]: classes %s fused %d | wall %.1f us/substep (graph) | events: surfaces %.1f us, substep %.1f us | "
          "algorithmic %.3f GB -> %.0f GB/s on the whole sub-timestep" % (
              cfg, mode, b.class_counts(), b.n_fused_surfaces, wall, su, ss, ab / 1e9, ab / wall / 1e3), flush=True)
