"""The *model dict*: the flattened description of a thermal model that both the C ABI
(``heat_batch_desc``, include/heat_amd.h) and the test oracle consume, plus

* ``layout_state``  — SimulationState slot numbering and initial values exactly as the
  reference registers them (zones first: src/zone.rs:45-50 via src/model.rs:225-230; then per
  surface 8 scalar slots and the node temperatures: src/surface.rs:428-442,
  src/surface_trait.rs:223-378);
* synthetic workloads for tests and bench (BASELINE.json configs 2, 3, 5 and the headline).

Pure numpy; no device code here.
"""
import math

import numpy as np

SPACE, AMBIENT, OUTDOOR, GROUND = 0, 1, 2, 3
AIR, ARGON, KRYPTON, XENON = 0, 1, 2, 3
SIGMA = 5.670374419e-8          # src/lib.rs:49
HS_INIT = 1.739658084820765     # src/surface_trait.rs:231,248
T_INIT = 22.0                   # src/surface_trait.rs:368, src/zone.rs:48
MAX_RS = 0.05                   # src/discretization.rs:21

CAVITY_DTYPE = np.dtype([("thickness", "f8"), ("height", "f8"), ("angle", "f8"), ("eout", "f8"),
                         ("ein", "f8"), ("gas", "i4"), ("pad_", "i4")])

PER_SURFACE_F64 = ["front_ambient", "back_ambient", "front_emissivity", "back_emissivity", "area", "perimeter",
                   "cos_tilt", "normal_x", "normal_y", "wind_modifier"]
PER_SURFACE_I32 = ["front_kind", "back_kind", "front_zone", "back_zone"]
SLOT_KEYS = ["hs_front_slot", "hs_back_slot", "flow_front_slot", "flow_back_slot", "solar_front_slot",
             "solar_back_slot", "ir_front_slot", "ir_back_slot"]


def wind_speed_modifier(height, terrain="none"):
    """src/surface.rs:135-166. terrain: 'none' (site_details = None -> Urban), or a TerrainClass name."""
    if height < 1e-5:
        return 0.0
    table = {"country": (0.14, 270.), "suburbs": (0.22, 370.), "city": (0.33, 460.), "ocean": (0.10, 210.),
             "urban": (0.22, 370.), "none": (0.22, 370.)}
    alpha, delta = table[terrain]
    return (270. / 10.) ** 0.14 * (height / delta) ** alpha


def layout_state(md):
    """Assigns SimulationState slots in the reference's registration order and returns the initial state."""
    S, Z = int(md["n_surfaces"]), int(md["n_zones"])
    off = np.asarray(md["node_offset"], dtype=np.int64)
    n = np.diff(off)
    md["zone_slot"] = np.arange(Z, dtype=np.int64)
    base = Z + np.concatenate(([0], np.cumsum(8 + n)))[:-1] if S else np.zeros(0, dtype=np.int64)
    base = base.astype(np.int64)
    for i, k in enumerate(SLOT_KEYS):
        md[k] = base + i
    md["first_node_slot"] = base + 8
    n_state = int(Z + (8 * S + n.sum() if S else 0))
    md["n_state"] = n_state
    return initial_state(md)


def initial_state(md):
    st = np.zeros(int(md["n_state"]), dtype=np.float64)
    st[md["zone_slot"]] = T_INIT
    st[md["hs_front_slot"]] = HS_INIT
    st[md["hs_back_slot"]] = HS_INIT
    off = np.asarray(md["node_offset"], dtype=np.int64)
    n = np.diff(off)
    first = np.asarray(md["first_node_slot"], dtype=np.int64)
    if len(n):
        idx = np.repeat(first - off[:-1], n) + np.arange(off[-1])
        st[idx] = T_INIT
    return st


def node_slots(md):
    """State slot of every node, in CSR order."""
    off = np.asarray(md["node_offset"], dtype=np.int64)
    n = np.diff(off)
    first = np.asarray(md["first_node_slot"], dtype=np.int64)
    return np.repeat(first - off[:-1], n) + np.arange(off[-1])


def empty(n_surfaces, n_zones, dt):
    md = dict(n_surfaces=int(n_surfaces), n_zones=int(n_zones), dt=float(dt), seg_cavity=None, cavities=None,
              front_hs_fix=None, back_hs_fix=None)
    return md


def subset(md, idx):
    """The model dict of surfaces ``idx`` only (zones and state slots unchanged): one rank's shard."""
    idx = np.asarray(idx, dtype=np.int64)
    off = np.asarray(md["node_offset"], dtype=np.int64)
    n = np.diff(off)[idx]
    new_off = np.concatenate(([0], np.cumsum(n))).astype(np.int64)
    node_idx = (np.repeat(off[idx] - new_off[:-1], n) + np.arange(new_off[-1])) if len(idx) else np.zeros(0, np.int64)
    out = dict(md)
    out["n_surfaces"] = len(idx)
    out["node_offset"] = new_off
    for k in ("mass", "uvalue", "front_alpha", "back_alpha"):
        out[k] = np.asarray(md[k])[node_idx]
    if md.get("seg_cavity") is not None:
        out["seg_cavity"] = np.asarray(md["seg_cavity"])[node_idx]
    for k in PER_SURFACE_F64 + PER_SURFACE_I32 + SLOT_KEYS + ["first_node_slot"]:
        out[k] = np.asarray(md[k])[idx]
    for k in ("front_hs_fix", "back_hs_fix"):
        if md.get(k) is not None:
            out[k] = np.asarray(md[k])[idx]
    return out


# ---------------------------------------------------------------------------
# Synthetic workloads
def weather_series(n_sub, dt, t0=0.0, wind_speed=3.0, wind_deg=150.0):
    """t_out(t) = 10 + 8 sin(2 pi t / 86400); constant wind. Returns [n_sub, 3] (C, radians, m/s)."""
    t = t0 + dt * (1 + np.arange(n_sub))
    w = np.empty((n_sub, 3))
    w[:, 0] = 10.0 + 8.0 * np.sin(2 * np.pi * t / 86400.0)
    w[:, 1] = math.radians(wind_deg)
    w[:, 2] = wind_speed
    return w


def _min_dx(dt_disc, k, rho_cp):
    """Positive root of the Euler-stability quadratic, src/discretization.rs:453-465."""
    b = -dt_disc / (rho_cp * MAX_RS)
    c = -2. * dt_disc * k / rho_cp
    return (-b + np.sqrt(b * b - 4. * c)) / 2.


def _fill_common(md, rng, S, Z, kinds_mode):
    """Geometry, boundaries, emissivities for S surfaces."""
    cos_choices = np.array([0.0, 1.0, -1.0, 0.707, -0.707])
    cos_tilt = cos_choices[rng.integers(0, 5, S)] if kinds_mode != "vertical" else np.zeros(S)
    az = rng.uniform(0, 2 * np.pi, S) if kinds_mode != "vertical" else np.full(S, -np.pi / 2)
    horiz = np.sqrt(np.maximum(0.0, 1.0 - cos_tilt ** 2))
    md["cos_tilt"] = cos_tilt
    md["normal_x"] = horiz * np.cos(az)
    md["normal_y"] = horiz * np.sin(az)
    if kinds_mode == "vertical":
        md["normal_x"] = np.zeros(S)
        md["normal_y"] = -np.ones(S)
    fk = np.full(S, OUTDOOR, dtype=np.int32)
    bk = np.full(S, SPACE, dtype=np.int32)
    if kinds_mode == "mixed":
        r = rng.random(S)
        ss = (r >= 0.80) & (r < 0.95)   # Space / Space
        ao = r >= 0.95                   # Ambient / Outdoor
        fk[ss] = SPACE
        fk[ao] = AMBIENT
        bk[ao] = OUTDOOR
    md["front_kind"], md["back_kind"] = fk, bk
    # surfaces of a zone are contiguous (the reference builds them space by space)
    zone_of = (np.arange(S, dtype=np.int64) * Z // max(S, 1)).astype(np.int32)
    md["back_zone"] = zone_of.copy()
    md["front_zone"] = ((zone_of + 1) % max(Z, 1)).astype(np.int32)  # Space/Space walls separate neighbours
    md["front_ambient"] = np.where(fk == AMBIENT, rng.uniform(5, 30, S), 0.0)
    md["back_ambient"] = np.zeros(S)
    md["zone_volume"] = rng.uniform(100., 600., Z) if kinds_mode != "vertical" else np.full(Z, 600.)
    if kinds_mode == "vertical":
        md["area"] = np.full(S, 60.0)
        md["perimeter"] = np.full(S, 46.0)
        md["wind_modifier"] = np.full(S, wind_speed_modifier(1.5))
        md["front_emissivity"] = np.full(S, 0.9)
        md["back_emissivity"] = np.full(S, 0.9)
    else:
        wdt = rng.uniform(2., 20., S)
        hgt = rng.uniform(2., 4., S)
        md["area"] = wdt * hgt
        md["perimeter"] = 2 * (wdt + hgt)
        zc = rng.uniform(0.5, 30., S)
        md["wind_modifier"] = (270. / 10.) ** 0.14 * (zc / 370.) ** 0.22
        md["front_emissivity"] = rng.uniform(0, 0.9, S)
        md["back_emissivity"] = rng.uniform(0, 0.9, S)


def _massive_nodes(n_nodes, k, rho_cp, dx):
    """mass / U of an all-massive single-material wall: n_nodes-1 elements of thickness dx
    (src/discretization.rs:190-220: each element gives half its mass to either end node)."""
    S = len(n_nodes)
    off = np.concatenate(([0], np.cumsum(n_nodes))).astype(np.int64)
    N = off[-1]
    surf = np.repeat(np.arange(S), n_nodes)
    local = np.arange(N) - off[surf]
    last = local == (n_nodes[surf] - 1)
    first = local == 0
    m_el = (rho_cp * dx)[surf]
    mass = np.where(first | last, m_el / 2., m_el / 2. + m_el / 2.)
    u = np.where(last, 0.0, (k / dx)[surf])
    return off, surf, local, first, last, mass, u


def uniform_massive(S, n, Z=None, dt=45.0, seed=20260401, identical=False, vertical=False):
    """S all-massive walls of n nodes each. ``identical`` + ``vertical`` = BASELINE config 2
    (concrete-like k=0.816, rho=1700, cp=800, dx=1/60 m; front Outdoor, back Space)."""
    rng = np.random.default_rng(seed)
    Z = Z if Z is not None else max(1, S // 100)
    md = empty(S, Z, dt)
    n_nodes = np.full(S, n, dtype=np.int64)
    if identical:
        k = np.full(S, 0.816)
        rho_cp = np.full(S, 1700. * 800.)
        dx = np.full(S, 1. / 60.)
    else:
        k, rho_cp, dx = _draw_materials(rng, S, dt)
    off, surf, local, first, last, mass, u = _massive_nodes(n_nodes, k, rho_cp, dx)
    md["node_offset"], md["mass"], md["uvalue"] = off, mass, u
    fa = np.zeros(off[-1]); ba = np.zeros(off[-1])
    fa[first] = 0.7
    ba[last] = 0.7
    md["front_alpha"], md["back_alpha"] = fa, ba
    _fill_common(md, rng, S, Z, "vertical" if vertical else "outdoor_space")
    state = layout_state(md)
    if not identical:
        perturb_initial_temperatures(md, state, rng)
        state[md["solar_front_slot"]] = rng.uniform(0, 800., S)
    set_ir_from_air(md, state, 10.0)
    return md, state


def partitioned_buildings(S, n, rooms=8, walls_per_room=12, partitions_per_room=2, dt=45.0, seed=20260401):
    """Buildings as real models have them: `rooms` zones per building, every room with `walls_per_room` all-massive
    walls of n nodes, `partitions_per_room` of which are interior partitions to the next room of the same building
    (front AND back face a Space, in different zones: src/model.rs:556-590 adds such a wall to both zones). The
    zones of a building form one cluster; buildings are independent of each other."""
    per_building = rooms * walls_per_room
    S = max(per_building, S // per_building * per_building)
    B = S // per_building
    Z = B * rooms
    md, state = uniform_massive(S, n, Z=Z, dt=dt, seed=seed)
    s = np.arange(S, dtype=np.int64)
    room = s // walls_per_room                       # global room (zone) number: rooms of a building are consecutive
    w = s % walls_per_room
    building = room // rooms
    nxt = building * rooms + (room % rooms + 1) % rooms
    part = w < partitions_per_room
    md["back_zone"] = room.astype(np.int32)
    md["front_zone"] = np.where(part, nxt, room).astype(np.int32)
    md["front_kind"] = np.where(part, SPACE, OUTDOOR).astype(np.int32)
    set_ir_from_air(md, state, 10.0)
    return md, state


def _draw_materials(rng, S, dt):
    """k in U[0.03,2], rho*cp in U[4e4,2.5e6], dx in U[0.005,0.04], re-drawn until the reference's
    Euler bound (discretization.rs:453-465, evaluated at 2*dt because the model halves it, model.rs:328-331) holds."""
    k = rng.uniform(0.03, 2.0, S)
    rho_cp = rng.uniform(4e4, 2.5e6, S)
    dx = rng.uniform(0.005, 0.04, S)
    for _ in range(200):
        bad = dx < _min_dx(2 * dt, k, rho_cp)
        nb = int(bad.sum())
        if nb == 0:
            break
        k[bad] = rng.uniform(0.03, 2.0, nb)
        rho_cp[bad] = rng.uniform(4e4, 2.5e6, nb)
        dx[bad] = rng.uniform(0.005, 0.04, nb)
    bad = dx < _min_dx(2 * dt, k, rho_cp)
    dx[bad] = 0.04
    rho_cp[bad] = 2.0e6
    k[bad] = 0.1
    return k, rho_cp, dx


def set_ir_from_air(md, state, t_out):
    """IR irradiance slots = sigma (T + 273.15)^4 of the air each side sees."""
    fk, bk = np.asarray(md["front_kind"]), np.asarray(md["back_kind"])
    tf = np.where(fk == OUTDOOR, t_out, T_INIT)
    tb = np.where(bk == OUTDOOR, t_out, T_INIT)
    state[md["ir_front_slot"]] = SIGMA * (tf + 273.15) ** 4
    state[md["ir_back_slot"]] = SIGMA * (tb + 273.15) ** 4


def ragged_mixed(S, Z=None, dt=45.0, seed=20260401, n_lo=8, n_hi=64):
    """BASELINE config 3: ragged node counts in [n_lo, n_hi]; 70 % all-massive, 20 % massive core
    between two no-mass facings (the polyurethane / concrete / polyurethane pattern), 10 % pure
    no-mass 2-node walls; mixed boundary kinds, tilts, emissivities and solar gains."""
    rng = np.random.default_rng(seed)
    Z = Z if Z is not None else max(1, S // 100)
    md = empty(S, Z, dt)
    kind = rng.random(S)
    pure = kind >= 0.9
    mixed = (kind >= 0.7) & ~pure
    n_nodes = rng.integers(n_lo, n_hi + 1, S).astype(np.int64)
    n_nodes[pure] = 2
    n_nodes[mixed] = np.maximum(n_nodes[mixed], 4)
    k, rho_cp, dx = _draw_materials(rng, S, dt)
    off, surf, local, first, last, mass, u = _massive_nodes(n_nodes, k, rho_cp, dx)
    # facings: node 0 and node n-1 carry no mass, their segment is a thin insulation layer
    u_ins = rng.uniform(0.5, 3.0, S)
    mx = mixed[surf]
    second = local == 1
    before_last = local == (n_nodes[surf] - 2)
    m_el = (rho_cp * dx)[surf]
    mass = np.where(mx & (first | last), 0.0, mass)
    mass = np.where(mx & (second | before_last), m_el / 2., mass)
    u = np.where(mx & first, u_ins[surf], u)
    u = np.where(mx & before_last, u_ins[surf], u)
    # pure no-mass: one thin layer
    pm = pure[surf]
    mass = np.where(pm, 0.0, mass)
    u = np.where(pm & first, rng.uniform(0.5, 5.0, S)[surf], u)
    md["node_offset"], md["mass"], md["uvalue"] = off, mass, u
    fa = np.zeros(off[-1]); ba = np.zeros(off[-1])
    fa[first] = rng.uniform(0.1, 0.9, S)
    ba[last] = rng.uniform(0.1, 0.9, S)
    md["front_alpha"], md["back_alpha"] = fa, ba
    _fill_common(md, rng, S, Z, "mixed")
    # The reference's no-mass update T <- (T + x(T))/2 applies long-wave radiation explicitly
    # (discretization.rs:663-664) and diverges from step to step once 4 eps sigma T^3 exceeds about
    # three times the other conductances of the node. Keep the no-mass faces in the convergent regime.
    light = mixed | pure
    md["front_emissivity"] = np.where(light, md["front_emissivity"] * (0.2 / 0.9), md["front_emissivity"])
    md["back_emissivity"] = np.where(light, md["back_emissivity"] * (0.2 / 0.9), md["back_emissivity"])
    state = layout_state(md)
    perturb_initial_temperatures(md, state, rng)
    state[md["solar_front_slot"]] = rng.uniform(0, 800., S)
    set_ir_from_air(md, state, 10.0)
    return md, state


def clustered_massive(S, Z=None, dt=45.0, seed=11, n_lo=8, n_hi=40, pair_fraction=0.12):
    """Ragged fast-path walls in small zone-connected clusters (the cluster-resident march's home ground):
    75 % all-massive, 25 % massive core between two no-mass facings; most walls Outdoor / Space, some
    Space / Space between the two zones of a pair (clusters of two zones), some Ambient / Space, some
    Ambient / Outdoor (coupled to no zone at all)."""
    rng = np.random.default_rng(seed)
    Z = Z if Z is not None else max(2, S // 25)
    md = empty(S, Z, dt)
    mixed = rng.random(S) >= 0.75
    n_nodes = rng.integers(n_lo, n_hi + 1, S).astype(np.int64)
    n_nodes[mixed] = np.maximum(n_nodes[mixed], 4)
    k, rho_cp, dx = _draw_materials(rng, S, dt)
    off, surf, local, first, last, mass, u = _massive_nodes(n_nodes, k, rho_cp, dx)
    u_ins = rng.uniform(0.5, 3.0, S)
    mx = mixed[surf]
    second = local == 1
    before_last = local == (n_nodes[surf] - 2)
    m_el = (rho_cp * dx)[surf]
    mass = np.where(mx & (first | last), 0.0, mass)
    mass = np.where(mx & (second | before_last), m_el / 2., mass)
    u = np.where(mx & first, u_ins[surf], u)
    u = np.where(mx & before_last, u_ins[surf], u)
    md["node_offset"], md["mass"], md["uvalue"] = off, mass, u
    fa = np.zeros(off[-1]); ba = np.zeros(off[-1])
    fa[first] = rng.uniform(0.1, 0.9, S)
    ba[last] = rng.uniform(0.1, 0.9, S)
    md["front_alpha"], md["back_alpha"] = fa, ba
    _fill_common(md, rng, S, Z, "outdoor_space")
    r = rng.random(S)
    zone_of = md["back_zone"].copy()
    fk, bk = md["front_kind"], md["back_kind"]
    pair = r < pair_fraction                      # Space / Space inside a pair of zones (2j, 2j+1)
    amb = (r >= pair_fraction) & (r < pair_fraction + 0.06)
    lone = r >= 0.95
    fk[pair] = SPACE
    md["front_zone"] = np.where(pair, np.minimum(zone_of ^ 1, Z - 1), zone_of).astype(np.int32)
    fk[amb | lone] = AMBIENT
    bk[lone] = OUTDOOR
    md["front_ambient"] = np.where(fk == AMBIENT, rng.uniform(5, 30, S), 0.0)
    md["front_emissivity"] = np.where(mixed, md["front_emissivity"] * (0.2 / 0.9), md["front_emissivity"])
    md["back_emissivity"] = np.where(mixed, md["back_emissivity"] * (0.2 / 0.9), md["back_emissivity"])
    state = layout_state(md)
    perturb_initial_temperatures(md, state, rng)
    state[md["solar_front_slot"]] = rng.uniform(0, 800., S)
    set_ir_from_air(md, state, 10.0)
    return md, state


def rooms_with_windows(S, Z=None, dt=45.0, seed=23, n_lo=8, n_hi=32, window_fraction=0.2, thin_fraction=0.08):
    """Rooms as real models have them: massive walls (some with no-mass facings) plus double-glazed windows
    (4 no-mass nodes around a gas cavity, src/cavity.rs:79-88) and thin no-mass partitions (2 nodes), all facing
    the room's zone; some walls and partitions separate the two zones of a pair. The clusters mix fast-path and
    small surfaces (the "mixed" workgroups of the cluster-resident march)."""
    rng = np.random.default_rng(seed)
    Z = Z if Z is not None else max(2, S // 20)
    md = empty(S, Z, dt)
    r = rng.random(S)
    glazing = r < window_fraction
    thin = (r >= window_fraction) & (r < window_fraction + thin_fraction)
    facing = (r >= window_fraction + thin_fraction) & (r < window_fraction + thin_fraction + 0.2)
    n_nodes = rng.integers(n_lo, n_hi + 1, S).astype(np.int64)
    n_nodes[glazing] = 4
    n_nodes[thin] = 2
    k, rho_cp, dx = _draw_materials(rng, S, dt)
    off, surf, local, first, last, mass, u = _massive_nodes(n_nodes, k, rho_cp, dx)
    # walls with no-mass facings (as in ragged_mixed)
    u_ins = rng.uniform(0.5, 3.0, S)
    fx = facing[surf]
    second = local == 1
    before_last = local == (n_nodes[surf] - 2)
    m_el = (rho_cp * dx)[surf]
    mass = np.where(fx & (first | last), 0.0, mass)
    mass = np.where(fx & (second | before_last), m_el / 2., mass)
    u = np.where(fx & first, u_ins[surf], u)
    u = np.where(fx & before_last, u_ins[surf], u)
    # thin no-mass partitions
    tn = thin[surf]
    mass = np.where(tn, 0.0, mass)
    u = np.where(tn & first, rng.uniform(0.5, 5.0, S)[surf], u)
    # double glazing
    gz = glazing[surf]
    mass = np.where(gz, 0.0, mass)
    u = np.where(gz & ((local == 0) | (local == 2)), 1.0 / 0.003, u)
    u = np.where(gz & ((local == 1) | (local == 3)), 0.0, u)
    segc = np.full(off[-1], -1, dtype=np.int32)
    g_idx = np.cumsum(glazing) - 1
    sel = gz & (local == 1)
    segc[sel] = g_idx[surf[sel]]
    tilt = rng.choice([math.pi / 2, math.pi / 2, math.radians(73.0), math.radians(30.0)], S)
    n_g = int(glazing.sum())
    cav = np.zeros(n_g, dtype=CAVITY_DTYPE)
    cav["thickness"], cav["height"], cav["angle"] = 0.0127, 1.0, tilt[glazing]
    cav["eout"], cav["ein"], cav["gas"] = 0.84, 0.84, rng.integers(0, 4, n_g)
    md["node_offset"], md["mass"], md["uvalue"] = off, mass, u
    if n_g:
        md["seg_cavity"], md["cavities"] = segc, cav
    fa = np.zeros(off[-1]); ba = np.zeros(off[-1])
    fa[first] = rng.uniform(0.1, 0.9, S)
    ba[last] = rng.uniform(0.1, 0.9, S)
    a_f = np.array([0.05, 0.05, 0.03, 0.03])
    fa = np.where(gz, a_f[np.minimum(local, 3)], fa)
    ba = np.where(gz, a_f[::-1][np.minimum(local, 3)], ba)
    md["front_alpha"], md["back_alpha"] = fa, ba
    _fill_common(md, rng, S, Z, "outdoor_space")
    md["cos_tilt"] = np.where(glazing, np.cos(tilt), md["cos_tilt"])
    rr = rng.random(S)
    zone_of = md["back_zone"].copy()
    fk = md["front_kind"]
    pair = (rr < 0.12) & ~glazing                # interior walls / partitions between the two zones of a pair
    fk[pair] = SPACE
    md["front_zone"] = np.where(pair, np.minimum(zone_of ^ 1, Z - 1), zone_of).astype(np.int32)
    light = facing | thin | glazing
    md["front_emissivity"] = np.where(light, md["front_emissivity"] * (0.2 / 0.9), md["front_emissivity"])
    md["back_emissivity"] = np.where(light, md["back_emissivity"] * (0.2 / 0.9), md["back_emissivity"])
    state = layout_state(md)
    perturb_initial_temperatures(md, state, rng)
    state[md["solar_front_slot"]] = rng.uniform(0, 800., S)
    state[md["solar_back_slot"]] = np.where(glazing, rng.uniform(0, 100., S), 0.0)
    set_ir_from_air(md, state, 10.0)
    return md, state


def perturb_initial_temperatures(md, state, rng):
    """Spreads the initial temperatures (the reference's all-22.0 start makes every natural
    convection coefficient hit its 0.1 floor, convection.rs:22,91-92)."""
    ns = node_slots(md)
    state[ns] = T_INIT + rng.uniform(-3.0, 3.0, len(ns))
    state[md["zone_slot"]] = T_INIT + rng.uniform(-2.0, 2.0, int(md["n_zones"]))


def glazing_cavity(S, Z=None, dt=45.0, seed=7, trombe_fraction=0.5):
    """BASELINE config 5: (i) double glazing — 3 mm glass / 12.7 mm air / 3 mm glass, 4 no-mass nodes,
    the cavity of src/cavity.rs:79-88, solar absorbed in every node (src/surface.rs:486-494);
    (ii) Trombe-like — concrete 0.2 m (12 elements) / 5 cm air / 3 cm glass (3 elements), all massive,
    cavity segment inside the chunk (tests/validate_wall_heat_transfer.rs:1095-1099)."""
    rng = np.random.default_rng(seed)
    Z = Z if Z is not None else max(1, S // 100)
    md = empty(S, Z, dt)
    trombe = rng.random(S) < trombe_fraction
    n_nodes = np.where(trombe, 17, 4).astype(np.int64)
    off = np.concatenate(([0], np.cumsum(n_nodes))).astype(np.int64)
    N = off[-1]
    mass = np.zeros(N); u = np.zeros(N); segc = np.full(N, -1, dtype=np.int32)
    fa = np.zeros(N); ba = np.zeros(N)
    cav = np.zeros(S, dtype=CAVITY_DTYPE)
    tilt = rng.choice([math.pi / 2, math.pi / 3, math.radians(73.0), math.radians(30.0), math.radians(134.0)], S)
    for s in range(S):
        o = off[s]
        if trombe[s]:
            dxc = 0.2 / 12
            mc = 1700. * 800. * dxc
            mass[o:o + 13] = mc
            mass[o] = mc / 2; mass[o + 12] = mc / 2
            u[o:o + 12] = 0.816 / dxc
            segc[o + 12] = s
            dxg = 0.03 / 3
            mg = 2500. * 840. * dxg
            mass[o + 13:o + 17] = mg
            mass[o + 13] = mg / 2; mass[o + 16] = mg / 2
            u[o + 13:o + 16] = 1.0 / dxg
            fa[o] = 0.7
            ba[o + 16] = 0.1
            cav[s] = (0.05, 1.0, tilt[s], 0.9, 0.84, AIR, 0)
        else:
            u[o] = 1.0 / 0.003
            segc[o + 1] = s
            u[o + 2] = 1.0 / 0.003
            a_f = np.array([0.05, 0.05, 0.03, 0.03])
            fa[o:o + 4] = a_f
            ba[o:o + 4] = a_f[::-1]
            cav[s] = (0.0127, 1.0, tilt[s], 0.84, 0.84, rng.integers(0, 4), 0)
    md["node_offset"], md["mass"], md["uvalue"] = off, mass, u
    md["seg_cavity"], md["cavities"] = segc, cav
    md["front_alpha"], md["back_alpha"] = fa, ba
    _fill_common(md, rng, S, Z, "outdoor_space")
    md["cos_tilt"] = np.cos(tilt)
    horiz = np.sin(tilt)
    az = rng.uniform(0, 2 * np.pi, S)
    md["normal_x"], md["normal_y"] = horiz * np.cos(az), horiz * np.sin(az)
    state = layout_state(md)
    state[md["solar_front_slot"]] = rng.uniform(0, 800., S)
    state[md["solar_back_slot"]] = rng.uniform(0, 800., S)
    set_ir_from_air(md, state, 10.0)
    return md, state
