"""Shared builders for tests: single-surface model dicts in the shape of the reference's unit tests."""
import math

import numpy as np

from heat_amd import modeldict as mdl

BRICKWORK = dict(k=0.816, rho=1700., cp=800., front_thermal_abs=0., back_thermal_abs=0.)   # surface.rs:1061-1075
POLYURETHANE = dict(k=0.0252, rho=17.5, cp=2400., front_thermal_abs=0., back_thermal_abs=0.)  # surface.rs:1048-1059
CONCRETE = dict(k=0.816, rho=1700., cp=800.)  # tests/massive_*/in.idf, tests/tilted/back.spl


def surfaces_model(segs, dt, front_kind, back_kind, n_zones=0, zone_volume=(), front_zone=0, back_zone=0,
                   front_ambient=0.0, back_ambient=0.0, front_emis=0.0, back_emis=0.0, area=4.0, perimeter=8.0,
                   cos_tilt=1.0, normal=(0., 0., 1.), wind_modifier=None, height=10.0, hs_fix=None, copies=1):
    """A model dict with `copies` identical surfaces built from one discretization dict `segs`
    (keys mass, uvalue, seg_cavity, cavities, front_alpha, back_alpha)."""
    n = len(segs["mass"])
    S = copies
    md = mdl.empty(S, n_zones, dt)
    md["node_offset"] = np.arange(S + 1, dtype=np.int64) * n
    for k in ("mass", "uvalue", "front_alpha", "back_alpha"):
        md[k] = np.tile(np.asarray(segs[k], dtype=np.float64), S)
    cav = segs.get("cavities")
    if cav is not None and len(cav):
        nc = len(cav)
        sc = np.asarray(segs["seg_cavity"], dtype=np.int32)
        allsc = []
        allcav = []
        for s in range(S):
            allsc.append(np.where(sc >= 0, sc + s * nc, -1))
            allcav.append(cav)
        md["seg_cavity"] = np.concatenate(allsc).astype(np.int32)
        md["cavities"] = np.concatenate(allcav)
    md["front_kind"] = np.full(S, front_kind, dtype=np.int32)
    md["back_kind"] = np.full(S, back_kind, dtype=np.int32)
    md["front_zone"] = np.full(S, front_zone, dtype=np.int32)
    md["back_zone"] = np.full(S, back_zone, dtype=np.int32)
    md["front_ambient"] = np.full(S, front_ambient)
    md["back_ambient"] = np.full(S, back_ambient)
    md["front_emissivity"] = np.full(S, front_emis)
    md["back_emissivity"] = np.full(S, back_emis)
    md["area"] = np.full(S, area)
    md["perimeter"] = np.full(S, perimeter)
    md["cos_tilt"] = np.full(S, cos_tilt)
    md["normal_x"] = np.full(S, normal[0])
    md["normal_y"] = np.full(S, normal[1])
    wm = mdl.wind_speed_modifier(height) if wind_modifier is None else wind_modifier
    md["wind_modifier"] = np.full(S, wm)
    if hs_fix is not None:
        md["front_hs_fix"] = np.full(S, hs_fix[0])
        md["back_hs_fix"] = np.full(S, hs_fix[1])
    md["zone_volume"] = np.asarray(zone_volume, dtype=np.float64)
    state = mdl.layout_state(md)
    return md, state


def random_zone_graph_model(seed):
    """The random small models of test_planner_stress_random_zone_graphs: walls, facings, windows, partitions, walls
    between random pairs of zones (longer chains), zones nobody faces, walls facing the same zone on both sides,
    walls facing no zone. Returns (model dict, initial state)."""
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.integers(60, 900))
    Z = int(rng.integers(3, 40))
    gen = mdl.rooms_with_windows if seed % 2 else mdl.clustered_massive
    md, st = gen(S, Z=Z, dt=45.0, seed=seed)
    pick = rng.random(S)
    both = (md["front_kind"] == mdl.SPACE) & (md["back_kind"] == mdl.SPACE)
    rew = both & (pick < 0.3)
    md["front_zone"] = np.where(rew, rng.integers(0, Z, S), md["front_zone"]).astype(np.int32)
    same = both & (pick > 0.9)
    md["front_zone"] = np.where(same, md["back_zone"], md["front_zone"]).astype(np.int32)
    nodes = np.diff(md["node_offset"])
    lone = (pick > 0.5) & (pick < 0.56) & (nodes > 4)
    md["front_kind"] = np.where(lone, mdl.AMBIENT, md["front_kind"]).astype(np.int32)
    md["back_kind"] = np.where(lone, mdl.OUTDOOR, md["back_kind"]).astype(np.int32)
    md["front_ambient"] = np.where(lone, 12.5, md["front_ambient"])
    md["_rng"] = rng
    return md, st


def walls_example_model(n_walls=12):
    """The building of examples/march_walls.cpp as a model dict, discretized by the oracle's restatement of the
    setup (so that it needs no device library): 12 walls — concrete / insulation-concrete-insulation / insulation —
    around two zones, every fifth wall between the zones."""
    from oracle import oracle as orc
    ins = dict(thickness=0.02, k=0.0252, rho=17.5, cp=2400.)
    conc = dict(thickness=0.2, k=0.816, rho=1700., cp=800.)
    mass, uval, offs = [], [], [0]
    main_dt = 3600.0 / 20
    segs = []
    for i in range(n_walls):
        layers = [conc] if i % 3 == 0 else ([ins, conc, ins] if i % 3 == 1 else [ins])
        segs.append(orc.discretize(layers, main_dt, 0.04, 60.0))
    dt = main_dt / max(sg["tstep_subdivision"] for sg in segs) / 2.0
    for sg in segs:
        mass.extend(sg["mass"]); uval.extend(sg["uvalue"]); offs.append(offs[-1] + len(sg["mass"]))
    S = n_walls
    md = mdl.empty(S, 2, dt)
    md["node_offset"] = np.asarray(offs, dtype=np.int64)
    md["mass"] = np.asarray(mass); md["uvalue"] = np.asarray(uval)
    N = offs[-1]
    fa = np.zeros(N); ba = np.zeros(N)
    fa[md["node_offset"][:-1]] = 0.7
    ba[md["node_offset"][1:] - 1] = 0.7
    md["front_alpha"], md["back_alpha"] = fa, ba
    i = np.arange(S)
    md["front_kind"] = np.where(i % 5 == 4, mdl.SPACE, mdl.OUTDOOR).astype(np.int32)
    md["back_kind"] = np.full(S, mdl.SPACE, dtype=np.int32)
    md["front_zone"] = np.ones(S, dtype=np.int32)
    md["back_zone"] = (i % 2).astype(np.int32)
    md["front_ambient"] = np.zeros(S); md["back_ambient"] = np.zeros(S)
    md["front_emissivity"] = np.where(i % 3 == 0, 0.9, 0.2); md["back_emissivity"] = md["front_emissivity"].copy()
    md["area"] = 10.0 + i
    md["perimeter"] = 2.0 * (md["area"] / 3.0 + 3.0)
    md["cos_tilt"] = np.zeros(S)
    md["normal_x"] = np.sin(0.4 * i); md["normal_y"] = np.cos(0.4 * i)
    md["wind_modifier"] = np.array([mdl.wind_speed_modifier(1.5 + 3.0 * (q % 4)) for q in range(S)])
    md["zone_volume"] = np.array([600.0, 250.0])
    state = mdl.layout_state(md)
    return md, state
