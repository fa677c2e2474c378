"""Where a STREAMED tile's time goes (diagnostic build with in-kernel stamps, -DHEAT_STREAM_STAMPS; the product library
carries none):   python tools/stream_phases.py [CONFIG]
Per fast tile (lane 0 stamps the shader clock): loads issued -> arrived | boundary terms + no-mass loop | RK4 | new
coefficients + contributions | stores acknowledged; light tiles (8 / 4 nodes per lane) and wide tiles (16) apart.
The stamps drain the memory counters at two places (after the loads, after the stores), which the product does not."""
import os, sys, types, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heat_amd import build as _hb
os.environ["HEAT_AMD_LIB"] = _hb.build_variant("sstamps", ["HEAT_STREAM_STAMPS"])
import numpy as np
import bench
from heat_amd import HeatBatch, modeldict as mdl, binding
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
md, st, _ = bench.build_config(cfg, types.SimpleNamespace(surfaces=1_000_000, nodes=32), 45.0, 20260401)
w = mdl.weather_series(20, float(md["dt"]))
with HeatBatch(md, use_graph=True, no_fusion=True) as b:
    b.upload_state(st)
    for _ in range(10):
        b.march_resident(w)
    b.synchronize()
    b.set_timing(True)
    b.march_resident(w); b.synchronize()
    us, ss, _ = b.get_timing()
    L = binding.load_library()
    nb = 65536
    buf = np.zeros(nb * 8, dtype=np.uint64)
    L.heat_debug_stamps.restype = C.c_int
    assert L.heat_debug_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(nb)) == 0
    s = buf.reshape(nb, 8).astype(np.float64)
    print("%s streamed, stamped build: surfaces %.1f us, sub-timestep %.1f us; classes %s" % (cfg, us, ss, b.class_counts()))
    order = (0, 1, 6, 7, 2, 3)
    names = ("loads issued -> arrived", "boundary terms, no-mass loop", "RK4", "new coefficients, contributions", "stores acknowledged")
    for part, lo in (("light (8 / 4 nodes per lane)", 0), ("wide (16 nodes per lane)", 32768)):
        x = s[lo:lo + 32768]
        x = x[x[:, 3] > 0]
        ok = np.ones(len(x), bool)
        for a, e in zip(order[:-1], order[1:]):
            ok &= x[:, e] >= x[:, a]
        x = x[ok]
        if len(x) == 0:
            continue
        whole = x[:, 3] - x[:, 0]
        print("%s: %d tiles stamped, whole tile median %.0f ticks (mean %.0f, p90 %.0f)" % (part, len(x), np.median(whole), whole.mean(), np.percentile(whole, 90)))
        for name, a, e in zip(names, order[:-1], order[1:]):
            d = x[:, e] - x[:, a]
            print("   %-34s median %7.0f  mean %7.0f  p90 %7.0f   (%4.1f %% of the tile's mean)" % (name, np.median(d), d.mean(), np.percentile(d, 90), 100 * d.mean() / whole.mean()))
