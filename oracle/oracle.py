"""ctypes binding of the CPU oracle (oracle/libheat_oracle.so).

TEST INFRASTRUCTURE ONLY: may be imported from tests/, from
``__graft_entry__.smoke()`` and from ``bench.py``'s ``cpu_baseline`` leg — never
from the product package ``heat_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libheat_oracle.so")

SPACE, AMBIENT, OUTDOOR, GROUND = 0, 1, 2, 3
AIR, ARGON, KRYPTON, XENON = 0, 1, 2, 3

_p = C.POINTER
_d = C.c_double
_dp = _p(C.c_double)
_i32p = _p(C.c_int32)
_i64p = _p(C.c_int64)
_ip = _p(C.c_int)


class Cavity(C.Structure):
    _fields_ = [("thickness", _d), ("height", _d), ("angle", _d), ("eout", _d), ("ein", _d),
                ("gas", C.c_int32), ("pad_", C.c_int32)]


CAVITY_DTYPE = np.dtype([("thickness", "f8"), ("height", "f8"), ("angle", "f8"), ("eout", "f8"),
                         ("ein", "f8"), ("gas", "i4"), ("pad_", "i4")])


class Layer(C.Structure):
    _fields_ = [("is_gas", C.c_int32), ("gas", C.c_int32), ("thickness", _d), ("k", _d), ("rho", _d),
                ("cp", _d), ("front_thermal_abs", _d), ("back_thermal_abs", _d), ("tau", _d),
                ("front_solar_abs", _d), ("back_solar_abs", _d)]


class Model(C.Structure):
    _fields_ = [
        ("n_surfaces", C.c_int64), ("n_zones", C.c_int64), ("n_cavities", C.c_int64), ("dt", _d),
        ("node_offset", _i64p), ("mass", _dp), ("uvalue", _dp), ("seg_cavity", _i32p),
        ("front_alpha", _dp), ("back_alpha", _dp), ("cavities", _p(Cavity)),
        ("front_kind", _i32p), ("back_kind", _i32p), ("front_zone", _i32p), ("back_zone", _i32p),
        ("front_ambient", _dp), ("back_ambient", _dp), ("front_emissivity", _dp), ("back_emissivity", _dp),
        ("area", _dp), ("perimeter", _dp), ("cos_tilt", _dp), ("normal_x", _dp), ("normal_y", _dp),
        ("wind_modifier", _dp), ("front_hs_fix", _dp), ("back_hs_fix", _dp),
        ("first_node_slot", _i64p), ("hs_front_slot", _i64p), ("hs_back_slot", _i64p),
        ("flow_front_slot", _i64p), ("flow_back_slot", _i64p), ("solar_front_slot", _i64p),
        ("solar_back_slot", _i64p), ("ir_front_slot", _i64p), ("ir_back_slot", _i64p),
        ("zone_volume", _dp), ("zone_slot", _i64p),
    ]


def build(force=False):
    """Compile oracle/heat_oracle.c with gcc (``make -C oracle``)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "heat_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.or_is_windward.restype = C.c_int
        L.or_is_windward.argtypes = [_d, _d, _d, _d]
        L.or_wind_speed_modifier.restype = _d
        L.or_wind_speed_modifier.argtypes = [_d, C.c_int, C.c_int]
        L.or_tarp_natural.restype = _d
        L.or_tarp_natural.argtypes = [_d, _d, _d, _ip]
        L.or_tarp_total.restype = _d
        L.or_tarp_total.argtypes = [_d, _d, _d, _d, _d, _d, C.c_int, _ip]
        for f in ("or_gas_thermal_conductivity", "or_gas_dynamic_viscosity", "or_gas_heat_capacity", "or_gas_density"):
            getattr(L, f).restype = _d
            getattr(L, f).argtypes = [C.c_int, _d]
        L.or_gas_mass.restype = _d
        L.or_gas_mass.argtypes = [C.c_int]
        L.or_raleigh.restype = _d
        L.or_raleigh.argtypes = [C.c_int, _d, _d, _d]
        L.or_nusselt.restype = _d
        L.or_nusselt.argtypes = [_d, _d, _d, _ip]
        L.or_cavity_convection.restype = _d
        L.or_cavity_convection.argtypes = [C.c_int, _d, _d, _d, _d, _d, _ip]
        L.or_cavity_u_value.restype = _d
        L.or_cavity_u_value.argtypes = [_p(Cavity), _d, _d, _ip]
        L.or_zone_mcp.restype = _d
        L.or_zone_mcp.argtypes = [_d, _d]
        L.or_prod_tri_diag.restype = None
        L.or_prod_tri_diag.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.or_tri_diag_gaussian.restype = None
        L.or_tri_diag_gaussian.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.or_rearrange_k.restype = None
        L.or_rearrange_k.argtypes = [C.c_int, _d, _dp, _dp, _dp, _dp, _dp]
        L.or_rk4.restype = None
        L.or_rk4.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.or_get_k_q.restype = C.c_int
        L.or_get_k_q.argtypes = [C.c_int, _dp, _i32p, _p(Cavity), C.c_int, C.c_int, _dp,
                                 _d, _d, _d, _d, _d, _d, _d, _d, _dp, _dp, _dp, _dp]
        L.or_get_chunks.restype = None
        L.or_get_chunks.argtypes = [C.c_int, _dp, _ip, _ip, _ip, _ip]
        L.or_model_march.restype = C.c_int
        L.or_model_march.argtypes = [_p(Model), _dp, _dp, C.c_int, _dp, _dp, _i64p]
        L.or_model_march_mt.restype = C.c_int
        L.or_model_march_mt.argtypes = [_p(Model), _dp, _dp, C.c_int, _dp, _dp, C.c_int]
        L.or_iterate_surfaces.restype = C.c_int
        L.or_iterate_surfaces.argtypes = [_p(Model), _dp, C.c_int64, C.c_int64, _d, _d, _d, _i64p]
        L.or_zones_abc.restype = None
        L.or_zones_abc.argtypes = [_p(Model), _dp, _dp, _dp, _dp]
        L.or_discretize_construction.restype = C.c_int
        L.or_discretize_construction.argtypes = [C.c_int, _p(Layer), _d, _d, _d, _ip]
        L.or_count_nodes.restype = C.c_int
        L.or_count_nodes.argtypes = [C.c_int, _ip]
        L.or_build.restype = C.c_int
        L.or_build.argtypes = [C.c_int, _p(Layer), _ip, _d, _d, _dp, _dp, _i32p, _p(Cavity), C.c_int]
        L.or_glazing_alphas.restype = C.c_int
        L.or_glazing_alphas.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
        L.or_glazing_combine_layers.restype = None
        L.or_glazing_combine_layers.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
        L.or_node_alphas.restype = C.c_int
        L.or_node_alphas.argtypes = [C.c_int, _p(Layer), _ip, C.c_int, _dp, _dp]
        _lib = L
    return _lib


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def dptr(a):
    return a.ctypes.data_as(_dp)


class OracleModel:
    """Holds an ``or_model`` built from the common model dict (see heat_amd.modeldict)."""

    _F64 = ["mass", "uvalue", "front_alpha", "back_alpha", "front_ambient", "back_ambient",
            "front_emissivity", "back_emissivity", "area", "perimeter", "cos_tilt", "normal_x",
            "normal_y", "wind_modifier", "zone_volume"]
    _I32 = ["front_kind", "back_kind", "front_zone", "back_zone"]
    _I64 = ["node_offset", "first_node_slot", "hs_front_slot", "hs_back_slot", "flow_front_slot",
            "flow_back_slot", "solar_front_slot", "solar_back_slot", "ir_front_slot", "ir_back_slot",
            "zone_slot"]

    def __init__(self, md):
        self.keep = {}
        m = Model()
        m.n_surfaces = int(md["n_surfaces"])
        m.n_zones = int(md["n_zones"])
        m.dt = float(md["dt"])
        for k in self._F64:
            a = _arr(md[k], np.float64)
            self.keep[k] = a
            setattr(m, k, a.ctypes.data_as(_dp))
        for k in self._I32:
            a = _arr(md[k], np.int32)
            self.keep[k] = a
            setattr(m, k, a.ctypes.data_as(_i32p))
        for k in self._I64:
            a = _arr(md[k], np.int64)
            self.keep[k] = a
            setattr(m, k, a.ctypes.data_as(_i64p))
        for k in ("front_hs_fix", "back_hs_fix"):
            if md.get(k) is not None:
                a = _arr(md[k], np.float64)
                self.keep[k] = a
                setattr(m, k, a.ctypes.data_as(_dp))
        if md.get("seg_cavity") is not None and md.get("cavities") is not None and len(md["cavities"]):
            sc = _arr(md["seg_cavity"], np.int32)
            cav = np.ascontiguousarray(md["cavities"], dtype=CAVITY_DTYPE)
            self.keep["seg_cavity"] = sc
            self.keep["cavities"] = cav
            m.seg_cavity = sc.ctypes.data_as(_i32p)
            m.cavities = cav.ctypes.data_as(_p(Cavity))
            m.n_cavities = len(cav)
        self.m = m
        self.n_state = int(md["n_state"])
        self.n_zones = m.n_zones

    def march(self, state, weather, zone_a0=None, zone_b0=None, threads=1):
        """In-place ThermalModel::march on ``state``; ``weather`` is [n_sub,3] (t_out, wind_dir_rad, wind_speed)."""
        assert state.dtype == np.float64 and state.flags.c_contiguous and state.size >= self.n_state
        w = _arr(weather, np.float64).reshape(-1, 3)
        a0 = _arr(zone_a0, np.float64) if zone_a0 is not None else None
        b0 = _arr(zone_b0, np.float64) if zone_b0 is not None else None
        iters = C.c_int64(0)
        if threads == 1:
            rc = lib().or_model_march(C.byref(self.m), dptr(state), dptr(w), len(w),
                                      dptr(a0) if a0 is not None else None,
                                      dptr(b0) if b0 is not None else None, C.byref(iters))
        else:
            rc = lib().or_model_march_mt(C.byref(self.m), dptr(state), dptr(w), len(w),
                                         dptr(a0) if a0 is not None else None,
                                         dptr(b0) if b0 is not None else None, int(threads))
        return rc, iters.value

    def iterate_surfaces(self, state, wind_direction, wind_speed, t_out, s0=0, s1=None):
        s1 = self.m.n_surfaces if s1 is None else s1
        iters = C.c_int64(0)
        rc = lib().or_iterate_surfaces(C.byref(self.m), dptr(state), s0, s1, wind_direction, wind_speed,
                                       t_out, C.byref(iters))
        return rc, iters.value

    def zones_abc(self, state, a0=None, b0=None):
        nz = self.m.n_zones
        a = np.zeros(nz) if a0 is None else np.array(a0, dtype=np.float64)
        b = np.zeros(nz) if b0 is None else np.array(b0, dtype=np.float64)
        c = np.zeros(nz)
        lib().or_zones_abc(C.byref(self.m), dptr(state), dptr(a), dptr(b), dptr(c))
        return a, b, c


def make_layers(layers):
    """layers: list of dicts -> ctypes array of or_layer (defaults as the reference applies them)."""
    arr = (Layer * len(layers))()
    for i, L in enumerate(layers):
        arr[i].is_gas = 1 if L.get("is_gas") else 0
        arr[i].gas = int(L.get("gas", AIR))
        arr[i].thickness = float(L["thickness"])
        arr[i].k = float(L.get("k", 0.0))
        arr[i].rho = float(L.get("rho", 0.0))
        arr[i].cp = float(L.get("cp", 0.0))
        arr[i].front_thermal_abs = float(L.get("front_thermal_abs", 0.84))
        arr[i].back_thermal_abs = float(L.get("back_thermal_abs", 0.84))
        arr[i].tau = float(L.get("tau", 0.0))
        arr[i].front_solar_abs = float(L.get("front_solar_abs", 0.84))
        arr[i].back_solar_abs = float(L.get("back_solar_abs", 0.84))
    return arr


def discretize(layers, model_dt, max_dx, min_dt, height=1.0, angle=0.0):
    """Discretization::new restated: returns dict(tstep_subdivision, n_elements, mass, uvalue, seg_cavity, cavities)."""
    L = lib()
    arr = make_layers(layers)
    n_layers = len(layers)
    n_el = (C.c_int * n_layers)()
    sub = L.or_discretize_construction(n_layers, arr, model_dt, max_dx, min_dt, n_el)
    return build_segments(layers, list(n_el), height, angle, tstep_subdivision=sub)


def build_segments(layers, n_elements, height=1.0, angle=0.0, tstep_subdivision=1):
    L = lib()
    arr = make_layers(layers)
    n_layers = len(layers)
    n_el = (C.c_int * n_layers)(*n_elements)
    n_nodes = L.or_count_nodes(n_layers, n_el)
    mass = np.zeros(n_nodes)
    uval = np.zeros(n_nodes)
    segc = np.zeros(n_nodes, dtype=np.int32)
    cav = np.zeros(max(n_layers, 1), dtype=CAVITY_DTYPE)
    nc = L.or_build(n_layers, arr, n_el, height, angle, dptr(mass), dptr(uval),
                    segc.ctypes.data_as(_i32p), cav.ctypes.data_as(_p(Cavity)), 0)
    if nc < 0:
        raise ValueError("or_build failed: %d" % nc)
    fa = np.zeros(n_nodes)
    ba = np.zeros(n_nodes)
    rc = L.or_node_alphas(n_layers, arr, n_el, n_nodes, dptr(fa), dptr(ba))
    return dict(tstep_subdivision=tstep_subdivision, n_elements=list(n_elements), n_nodes=n_nodes, mass=mass,
                uvalue=uval, seg_cavity=segc, cavities=cav[:nc].copy(), front_alpha=fa, back_alpha=ba,
                alpha_rc=rc)


MAX_NODES = 1024  # OR_MAX_NODES of heat_oracle.c: its scratch arrays per surface


def get_chunks(mass):
    mass = _arr(mass, np.float64)
    n = len(mass)
    if n > MAX_NODES:
        raise ValueError("the oracle holds %d nodes per surface at most (OR_MAX_NODES), got %d" % (MAX_NODES, n))
    nm, nn = C.c_int(0), C.c_int(0)
    mc = (C.c_int * (2 * n + 2))()
    nc = (C.c_int * (2 * n + 2))()
    lib().or_get_chunks(n, dptr(mass), C.byref(nm), mc, C.byref(nn), nc)
    return ([(mc[2 * i], mc[2 * i + 1]) for i in range(nm.value)],
            [(nc[2 * i], nc[2 * i + 1]) for i in range(nn.value)])
