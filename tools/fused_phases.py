"""Where a cluster-resident workgroup spends its time, and the clock the chip holds meanwhile (diagnostic build with
in-kernel stamps, heat_amd/build.py build_stamps; the product library carries none):
    python tools/fused_phases.py [CONFIG] [P]
Per FusedBlock: shader-clock ticks of init (loads, palette decode, zone data -> LDS), of the P sub-timesteps, and of the
final write-back; in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz over the block's lifetime."""
import os, sys, types, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heat_amd import build as _hb
os.environ["HEAT_AMD_LIB"] = _hb.build_stamps()  # (built on demand: __graft_entry__.build() treats it as optional)
import numpy as np
import bench
from heat_amd import HeatBatch, modeldict as mdl, binding
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 20
md, st, _ = bench.build_config(cfg, types.SimpleNamespace(surfaces=1_000_000, nodes=32), 45.0, 20260401)
w = mdl.weather_series(P, float(md["dt"]))
with HeatBatch(md, use_graph=True) as b:
    b.upload_state(st)
    for _ in range(60):          # the clock the chip settles at under this load, not the one it starts with
        b.march_resident(w)
    b.synchronize()
    b.set_timing(True)
    b.march_resident(w); b.synchronize()
    us, _, _ = b.get_timing()
    L = binding.load_library()
    nb = 65536
    buf = np.zeros(nb * 8, dtype=np.uint64)
    L.heat_debug_stamps.restype = C.c_int
    assert L.heat_debug_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(nb)) == 0
    s = buf.reshape(nb, 8)
    s = s[s[:, 3] > 0].astype(np.float64)
    ok = (s[:, 1] > s[:, 0]) & (s[:, 2] > s[:, 1]) & (s[:, 3] > s[:, 2]) & (s[:, 5] > s[:, 4])
    s = s[ok]  # (a slot is indexed by first tile / 4: workgroups of fewer than four tiles can share one — dropped where they mixed)
    clk = (s[:, 3] - s[:, 0]) / np.maximum(s[:, 5] - s[:, 4], 1.0) * 100e6
    print("%s, %d sub-timesteps per launch: %d stamped workgroups, launch %.1f us (%.2f us per sub-timestep)"
          % (cfg, P, len(s), us * P, us))
    print("in-kernel clock: median %.3f GHz (p10 %.3f, p90 %.3f)" % (np.median(clk) / 1e9, np.percentile(clk, 10) / 1e9, np.percentile(clk, 90) / 1e9))
    whole = s[:, 3] - s[:, 0]
    if (s[:, 6] > s[:, 0]).all() and (s[:, 7] >= s[:, 6]).all() and (s[:, 1] >= s[:, 7]).all():
        for name, a, e in (("  init: loads arrived", 0, 6), ("  init: palettes decoded", 6, 7), ("  init: zone data, barrier", 7, 1)):
            x = s[:, e] - s[:, a]
            print("%-28s median %9.0f ticks" % (name, np.median(x)))
    for name, a, e in (("init", 0, 1), ("sub-timesteps", 1, 2), ("write-back", 2, 3), ("whole", 0, 3)):
        x = s[:, e] - s[:, a]
        print("%-14s median %9.0f ticks = %6.2f us (%5.1f %% of the workgroup's life)%s" % (
            name, np.median(x), np.median(x) / np.median(clk) * 1e6, 100.0 * np.median(x) / np.median(whole),
            "   %.0f ticks per sub-timestep" % (np.median(x) / P) if name == "sub-timesteps" else ""))
