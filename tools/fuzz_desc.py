"""Malformed descriptors through the host-only planner under ASan / UBSan: a valid small model with one field damaged (an index
out of range, a negative count or slot, NaN / infinite / negative physical values, a Ground boundary, node offsets out of
order ...). The library must answer with an error code or plan the model — never crash, never read outside its arrays
(the reference panics or returns Err there: model.rs:88-92, surface.rs:642,687, discretization.rs:53). No GPU.
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/fuzz_desc.py [SECONDS] [FIRST_SEED]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from heat_amd import binding, build as hb, modeldict as mdl

L = binding.load_host_library(hb.build_plan_host())
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
INT_FIELDS = ["node_offset", "front_kind", "back_kind", "front_zone", "back_zone", "first_node_slot", "hs_front_slot", "hs_back_slot",
              "flow_front_slot", "flow_back_slot", "solar_front_slot", "solar_back_slot", "ir_front_slot", "ir_back_slot", "zone_slot", "seg_cavity"]
FLT_FIELDS = ["mass", "uvalue", "front_alpha", "back_alpha", "front_ambient", "back_ambient", "front_emissivity", "back_emissivity",
              "area", "perimeter", "cos_tilt", "normal_x", "normal_y", "wind_modifier", "zone_volume"]
n_err = n_ok = 0
gens = [lambda s: mdl.rooms_with_windows(90, Z=6, dt=45.0, seed=s), lambda s: mdl.ragged_mixed(70, Z=3, dt=45.0, seed=s),
        lambda s: mdl.glazing_cavity(60, Z=2, dt=45.0, seed=s), lambda s: mdl.partitioned_buildings(96, 9, rooms=4, dt=45.0, seed=s)]
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    md, _ = gens[seed % len(gens)](seed)
    for _ in range(int(rng.integers(1, 4))):
        if rng.random() < 0.55:
            k = INT_FIELDS[int(rng.integers(0, len(INT_FIELDS)))]
            if md.get(k) is None or len(md[k]) == 0:
                continue
            a = np.array(md[k]).copy()
            a[int(rng.integers(0, len(a)))] = int(rng.choice([-1, -7, 3, 2**31 - 1, 10**9, 0, 65, 4]))
            md[k] = a.astype(np.asarray(md[k]).dtype)
        elif rng.random() < 0.8:
            k = FLT_FIELDS[int(rng.integers(0, len(FLT_FIELDS)))]
            if md.get(k) is None or len(md[k]) == 0:
                continue
            a = np.array(md[k], dtype=np.float64).copy()
            a[int(rng.integers(0, len(a)))] = float(rng.choice([np.nan, np.inf, -np.inf, -1.0, 0.0, 1e300, -1e300, 1e-300]))
            md[k] = a
        elif rng.random() < 0.5:
            k = str(rng.choice(["n_state", "dt", "n_zones"]))
            if k == "dt":
                md[k] = [0, -1, 1, float("nan"), 1e300][int(rng.integers(0, 5))]
            elif k == "n_state":
                md[k] = int(rng.choice([0, -1, 1, 5]))
            else:  # fewer zones than the surfaces name (the zone arrays keep their length)
                md[k] = int(rng.choice([0, 1, max(0, int(md["n_zones"]) - 1)]))
        else:
            cav = md.get("cavities")
            if cav is not None and len(cav):  # a gas cavity's record
                f = str(rng.choice(["thickness", "height", "angle", "eout", "ein", "gas"]))
                cav = cav.copy()
                cav[f][int(rng.integers(0, len(cav)))] = (int(rng.choice([-1, 7, 2**31 - 1])) if f == "gas"
                                                          else float(rng.choice([np.nan, 0.0, -1.0, np.inf, 1e300])))
                md["cavities"] = cav
    opts = [dict(), dict(fuse_always=True), dict(no_fusion=True), dict(nodes_per_lane=int(rng.choice([4, 8, 16, 5, -4, 64]))), dict(force_general=True),
            dict(n_ranks=int(rng.choice([2, 0, -1])), rank=int(rng.choice([0, 1, 5, -1]))), dict(no_palette=True)][int(rng.integers(0, 7))]
    if os.environ.get("FUZZ_DESC_VERBOSE"):
        print("seed", seed, opts, flush=True)
    try:
        binding.plan_check(md, lib=L, **opts)
        n_ok += 1
    except binding.HeatError:
        n_err += 1
    if seed % 5 == 0:  # heat_partition on the damaged model, with sensible and senseless rank counts
        try:
            binding.partition(md, int(rng.choice([1, 2, 3, 8, 0, -2, 100000])), lib=L)
        except binding.HeatError:
            pass
    seed += 1
print("fuzz_desc: %d damaged descriptors refused with an error code, %d planned (the damage was harmless), no crash; seeds up to %d" % (n_err, n_ok, seed - 1))
