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
