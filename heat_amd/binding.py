"""ctypes binding of the C ABI in include/heat_amd.h.

``HeatBatch`` mirrors, at batch level, the reference's ``ThermalModel`` contract
(src/model.rs:188-428): ``HeatBatch(md)`` ≙ ``ThermalModel::new`` + ``allocate_memory``,
``HeatBatch.march(state, weather)`` ≙ ``ThermalModel::march``. Errors become ``HeatError``
(the reference returns ``Err(String)`` or panics).

The library is loaded from ``heat_amd/lib/libheat_amd.so``. If it is missing this module
raises: the HIP path is the only path.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_d = C.c_double
_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


class HeatError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("heat_amd error %d: %s" % (code, message))
        self.code = code
        self.message = message


class Cavity(C.Structure):
    _fields_ = [("thickness", _d), ("height", _d), ("angle", _d), ("eout", _d), ("ein", _d),
                ("gas", C.c_int32), ("reserved", C.c_int32)]


CAVITY_DTYPE = np.dtype([("thickness", "f8"), ("height", "f8"), ("angle", "f8"), ("eout", "f8"),
                         ("ein", "f8"), ("gas", "i4"), ("reserved", "i4")])


class Weather(C.Structure):
    _fields_ = [("dry_bulb", _d), ("wind_direction", _d), ("wind_speed", _d)]


class Desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("reserved", C.c_int32),
        ("n_surfaces", C.c_int64), ("n_zones", C.c_int64), ("n_cavities", C.c_int64), ("n_state", C.c_int64),
        ("dt", _d),
        ("node_offset", _i64p), ("mass", _dp), ("uvalue", _dp), ("seg_cavity", _i32p),
        ("front_alpha", _dp), ("back_alpha", _dp), ("cavities", C.POINTER(Cavity)),
        ("front_kind", _i32p), ("back_kind", _i32p), ("front_zone", _i32p), ("back_zone", _i32p),
        ("front_ambient", _dp), ("back_ambient", _dp), ("front_emissivity", _dp), ("back_emissivity", _dp),
        ("area", _dp), ("perimeter", _dp), ("cos_tilt", _dp), ("normal_x", _dp), ("normal_y", _dp),
        ("wind_modifier", _dp), ("front_hs_fix", _dp), ("back_hs_fix", _dp),
        ("first_node_slot", _i64p), ("hs_front_slot", _i64p), ("hs_back_slot", _i64p),
        ("flow_front_slot", _i64p), ("flow_back_slot", _i64p), ("solar_front_slot", _i64p),
        ("solar_back_slot", _i64p), ("ir_front_slot", _i64p), ("ir_back_slot", _i64p),
        ("zone_volume", _dp), ("zone_slot", _i64p),
    ]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("force_general", C.c_int32), ("nodes_per_lane", C.c_int32),
                ("use_graph", C.c_int32), ("stream", C.c_void_p), ("n_ranks", C.c_int32), ("rank", C.c_int32),
                ("no_palette", C.c_int32), ("no_fusion", C.c_int32)]


class Layer(C.Structure):
    """heat_layer (include/heat_amd_setup.h)"""
    _fields_ = [("is_gas", C.c_int32), ("gas", C.c_int32), ("thickness", _d), ("conductivity", _d), ("density", _d),
                ("specific_heat", _d), ("front_thermal_absorbtance", _d), ("back_thermal_absorbtance", _d),
                ("solar_transmittance", _d), ("front_solar_absorbtance", _d), ("back_solar_absorbtance", _d)]


class SurfaceIn(C.Structure):
    """heat_surface_in (include/heat_amd_setup.h)"""
    _fields_ = [("layers", C.POINTER(Layer)), ("n_layers", C.c_int32), ("is_fenestration", C.c_int32),
                ("area", _d), ("perimeter", _d), ("normal", _d * 3), ("centroid_z", _d),
                ("front_kind", C.c_int32), ("back_kind", C.c_int32), ("front_zone", C.c_int32),
                ("back_zone", C.c_int32), ("front_ambient", _d), ("back_ambient", _d)]


# Every symbol include/heat_amd.h and include/heat_amd_setup.h declare: (name, restype, argtypes)
_H = C.c_void_p
SYMBOLS = [
    ("heat_batch_create", C.c_int, [C.POINTER(Desc), C.POINTER(_H)]),
    ("heat_batch_create_ex", C.c_int, [C.POINTER(Desc), C.POINTER(Options), C.POINTER(_H)]),
    ("heat_batch_destroy", None, [_H]),
    ("heat_batch_upload_state", C.c_int, [_H, _dp, C.c_size_t]),
    ("heat_batch_download_state", C.c_int, [_H, _dp, C.c_size_t]),
    ("heat_batch_upload_inputs", C.c_int, [_H, _dp, C.c_size_t]),
    ("heat_batch_march", C.c_int, [_H, _dp, C.c_size_t, C.POINTER(Weather), C.c_int32, _dp, _dp]),
    ("heat_batch_march_ex", C.c_int, [_H, _dp, C.c_size_t, C.POINTER(Weather), C.c_int32, _dp, _dp, C.c_int32]),
    ("heat_batch_download_outputs", C.c_int, [_H, _dp, C.c_size_t, C.c_int32]),
    ("heat_batch_march_resident", C.c_int, [_H, C.POINTER(Weather), C.c_int32, _dp, _dp]),
    ("heat_batch_synchronize", C.c_int, [_H]),
    ("heat_batch_failed_surface", C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    ("heat_batch_set_weather", C.c_int, [_H, C.POINTER(Weather), C.c_int32, _dp, _dp]),
    ("heat_batch_step_surfaces", C.c_int, [_H, C.c_int32]),
    ("heat_batch_step_zones", C.c_int, [_H, C.c_void_p, C.c_int32]),
    ("heat_batch_zone_partials", C.c_void_p, [_H]),
    ("heat_batch_use_partials", C.c_int, [_H, C.c_void_p]),
    ("heat_batch_touched_zones", C.c_int, [_H, C.POINTER(C.c_uint8)]),
    ("heat_batch_set_shared_zones", C.c_int, [_H, _i32p, C.c_int32]),
    ("heat_comm_available", C.c_int, []),
    ("heat_comm_unique_id", C.c_int, [C.POINTER(C.c_uint8)]),
    ("heat_batch_comm_init", C.c_int, [_H, C.POINTER(C.c_uint8)]),
    ("heat_batch_comm_init_ex", C.c_int, [_H, C.POINTER(C.c_uint8), _i32p, C.c_int32]),
    ("heat_batch_set_owned_zones", C.c_int, [_H, C.POINTER(C.c_uint8)]),
    ("heat_batch_n_shared_zones", C.c_int32, [_H]),
    ("heat_batch_comm_ranks", C.c_int32, [_H]),
    ("heat_batch_comm_destroy", C.c_int, [_H]),
    ("heat_batch_set_fusion", C.c_int, [_H, C.c_int32]),
    ("heat_batch_n_fused_surfaces", C.c_int64, [_H]),
    ("heat_batch_n_fused_launches", C.c_int64, [_H]),
    ("heat_batch_n_surfaces", C.c_int64, [_H]),
    ("heat_batch_n_nodes", C.c_int64, [_H]),
    ("heat_batch_n_zones", C.c_int64, [_H]),
    ("heat_batch_algorithmic_bytes", C.c_int64, [_H]),
    ("heat_batch_nomass_iterations", C.c_int64, [_H]),
    ("heat_batch_class_counts", C.c_int, [_H, _i64p]),
    ("heat_batch_set_timing", C.c_int, [_H, C.c_int32]),
    ("heat_batch_get_timing", C.c_int, [_H, _dp, _dp, _i64p]),
    ("heat_partition", C.c_int, [C.POINTER(Desc), C.c_int32, _i32p, _i64p]),
    ("heat_batch_create_shard", C.c_int, [C.POINTER(Desc), C.POINTER(Options), _i32p, C.POINTER(_H)]),
    ("heat_plan_check", C.c_int, [C.POINTER(Desc), C.POINTER(Options), _i64p]),
    ("heat_last_error", C.c_char_p, []),
    ("heat_amd_abi_version", C.c_int, []),
    # include/heat_amd_setup.h
    ("heat_discretize_construction", C.c_int, [C.c_int32, C.POINTER(Layer), _d, _d, _d, _i32p]),
    ("heat_count_nodes", C.c_int32, [C.c_int32, _i32p]),
    ("heat_build_segments", C.c_int, [C.c_int32, C.POINTER(Layer), _i32p, _d, _d, _dp, _dp, _i32p,
                                      C.POINTER(Cavity), C.c_int32]),
    ("heat_get_chunks", C.c_int, [C.c_int32, _dp, _i32p, _i32p, _i32p, _i32p]),
    ("heat_glazing_alphas", C.c_int, [C.c_int32, _dp, _dp, _dp, _dp]),
    ("heat_node_alphas", C.c_int, [C.c_int32, C.POINTER(Layer), _i32p, C.c_int32, _dp, _dp]),
    ("heat_wind_speed_modifier", _d, [_d, C.c_int32]),
    ("heat_model_builder_create", C.c_void_p, [C.c_int32, C.c_int32]),
    ("heat_model_builder_destroy", None, [C.c_void_p]),
    ("heat_model_builder_add_zone", C.c_int, [C.c_void_p, _d]),
    ("heat_model_builder_add_surface", C.c_int, [C.c_void_p, C.POINTER(SurfaceIn)]),
    ("heat_model_builder_finish", C.c_int, [C.c_void_p, C.POINTER(C.POINTER(Desc)), C.POINTER(_dp), _i32p]),
    ("heat_model_builder_surface_info", C.c_int, [C.c_void_p, C.c_int64, _i32p, _i32p, _i32p, C.c_int32]),
]

_lib = None


def lib_path():
    # HEAT_AMD_LIB: another build of the same sources (measurement variants); the default is the in-tree library
    return os.environ.get("HEAT_AMD_LIB") or _build.LIB


def build_library(force=False):
    return _build.build(force=force)


def load_library():
    """Loads libheat_amd.so and binds every declared symbol. Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "heat_amd: %s is missing. Build it with `python -m heat_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % path)
    L = C.CDLL(path)
    for name, res, args in SYMBOLS:
        f = getattr(L, name)  # AttributeError if the library does not export it
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise HeatError(rc, load_library().heat_last_error().decode("utf-8", "replace"))


def as_weather(weather):
    """[n,3] array (dry bulb C, wind direction RADIANS, wind speed m/s) -> ctypes array."""
    w = np.ascontiguousarray(weather, dtype=np.float64).reshape(-1, 3)
    arr = (Weather * len(w))()
    if len(w):
        C.memmove(arr, w.ctypes.data, w.nbytes)
    return arr, len(w)


_F64 = ["mass", "uvalue", "front_alpha", "back_alpha", "front_ambient", "back_ambient", "front_emissivity",
        "back_emissivity", "area", "perimeter", "cos_tilt", "normal_x", "normal_y", "wind_modifier", "zone_volume"]
_I32 = ["front_kind", "back_kind", "front_zone", "back_zone"]
_I64 = ["node_offset", "first_node_slot", "hs_front_slot", "hs_back_slot", "flow_front_slot", "flow_back_slot",
        "solar_front_slot", "solar_back_slot", "ir_front_slot", "ir_back_slot", "zone_slot"]


def make_desc(md):
    """Builds a heat_batch_desc from the model dict (heat_amd.modeldict). Returns (desc, keepalive)."""
    keep = {}
    d = Desc()
    d.abi_version = 1
    d.n_surfaces = int(md["n_surfaces"])
    d.n_zones = int(md["n_zones"])
    d.n_state = int(md["n_state"])
    d.dt = float(md["dt"])
    for k in _F64:
        a = np.ascontiguousarray(md[k], dtype=np.float64)
        keep[k] = a
        setattr(d, k, a.ctypes.data_as(_dp))
    for k in _I32:
        a = np.ascontiguousarray(md[k], dtype=np.int32)
        keep[k] = a
        setattr(d, k, a.ctypes.data_as(_i32p))
    for k in _I64:
        a = np.ascontiguousarray(md[k], dtype=np.int64)
        keep[k] = a
        setattr(d, k, a.ctypes.data_as(_i64p))
    if md.get("front_hs_fix") is not None:
        for k in ("front_hs_fix", "back_hs_fix"):
            a = np.ascontiguousarray(md[k], dtype=np.float64)
            keep[k] = a
            setattr(d, k, a.ctypes.data_as(_dp))
    cav = md.get("cavities")
    if cav is not None and len(cav) and md.get("seg_cavity") is not None:
        sc = np.ascontiguousarray(md["seg_cavity"], dtype=np.int32)
        cv = np.zeros(len(cav), dtype=CAVITY_DTYPE)
        for f in ("thickness", "height", "angle", "eout", "ein", "gas"):
            cv[f] = cav[f]
        keep["seg_cavity"] = sc
        keep["cavities"] = cv
        d.seg_cavity = sc.ctypes.data_as(_i32p)
        d.cavities = cv.ctypes.data_as(C.POINTER(Cavity))
        d.n_cavities = len(cv)
    return d, keep


HOST_ONLY_SYMBOLS = ("heat_partition", "heat_plan_check", "heat_last_error", "heat_amd_abi_version")


def load_host_library(path):
    """Binds the host-only entry points (csrc/plan.cpp) of a library built without HIP: the sanitizer build of the
    planner that tests/test_planner_host.py runs in a child process."""
    L = C.CDLL(path)
    for name, res, args in SYMBOLS:
        if name in HOST_ONLY_SYMBOLS:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
    return L


def make_options(device=-1, force_general=False, nodes_per_lane=0, use_graph=False, stream=None, n_ranks=1, rank=0,
                 no_palette=False, no_fusion=False, fuse_always=False):
    opt = Options()
    opt.device = device
    opt.force_general = 1 if force_general else 0
    opt.nodes_per_lane = nodes_per_lane
    opt.use_graph = 1 if use_graph else 0
    opt.stream = stream
    opt.n_ranks = n_ranks
    opt.rank = rank
    opt.no_palette = 1 if no_palette else 0
    opt.no_fusion = 1 if no_fusion else (2 if fuse_always else 0)
    return opt


def partition(md, n_ranks, lib=None):
    """heat_partition: (rank of every surface, number of zones shared between ranks). Host-only."""
    L = lib or load_library()
    desc, keep = make_desc(md)
    ranks = np.zeros(int(md["n_surfaces"]), dtype=np.int32)
    n_shared = C.c_int64(0)
    rc = L.heat_partition(C.byref(desc), n_ranks, ranks.ctypes.data_as(_i32p), C.byref(n_shared))
    if rc != 0:
        raise HeatError(rc, L.heat_last_error().decode("utf-8", "replace"))
    return ranks, int(n_shared.value)


def plan_check(md, lib=None, **opts):
    """heat_plan_check: plans the model as heat_batch_create_ex would and verifies the plan. Returns the summary
    (surfaces per class [5], fused surfaces, fused workgroups, tiles). Host-only."""
    L = lib or load_library()
    desc, keep = make_desc(md)
    opt = make_options(**opts)
    summary = (C.c_int64 * 8)()
    rc = L.heat_plan_check(C.byref(desc), C.byref(opt), summary)
    if rc != 0:
        raise HeatError(rc, L.heat_last_error().decode("utf-8", "replace"))
    return list(summary)


def comm_available():
    """Whether the library can load RCCL (no collective inside: vote on it before comm_init)."""
    return load_library().heat_comm_available() == 0


def comm_unique_id():
    """ncclGetUniqueId through the library (128 bytes). One rank calls it and hands the bytes to the others."""
    buf = (C.c_uint8 * 128)()
    _check(load_library().heat_comm_unique_id(buf))
    return bytes(buf)


class HeatBatch:
    """Device-resident batch of surfaces + zones (≙ ThermalModel, src/model.rs:54-77)."""

    def __init__(self, md, device=-1, force_general=False, nodes_per_lane=0, use_graph=False, stream=None,
                 n_ranks=1, rank=0, no_palette=False, no_fusion=False, fuse_always=False, rank_of_surface=None):
        """rank_of_surface (heat_partition's result): the batch holds the surfaces of `rank` only, picked from the
        whole model's dict by the library (heat_batch_create_shard)."""
        self._L = load_library()
        self._h = _H()
        desc, keep = make_desc(md)
        opt = make_options(device, force_general, nodes_per_lane, use_graph, stream, n_ranks, rank, no_palette,
                           no_fusion, fuse_always)
        if rank_of_surface is None:
            _check(self._L.heat_batch_create_ex(C.byref(desc), C.byref(opt), C.byref(self._h)))
        else:
            ros = np.ascontiguousarray(rank_of_surface, dtype=np.int32)
            assert len(ros) == int(md["n_surfaces"])
            _check(self._L.heat_batch_create_shard(C.byref(desc), C.byref(opt), ros.ctypes.data_as(_i32p),
                                                   C.byref(self._h)))
        self.n_state = int(md["n_state"])
        self.n_zones = int(md["n_zones"])
        self.n_surfaces = int(md["n_surfaces"])

    def close(self):
        if getattr(self, "_h", None):
            self._L.heat_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @staticmethod
    def _state_ptr(state):
        assert state.dtype == np.float64 and state.flags.c_contiguous
        return state.ctypes.data_as(_dp)

    @staticmethod
    def _opt(a):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float64)
        return a, a.ctypes.data_as(_dp)

    def upload_state(self, state):
        _check(self._L.heat_batch_upload_state(self._h, self._state_ptr(state), state.size))

    def upload_inputs(self, state):
        _check(self._L.heat_batch_upload_inputs(self._h, self._state_ptr(state), state.size))

    def download_state(self, state):
        _check(self._L.heat_batch_download_state(self._h, self._state_ptr(state), state.size))

    OUT_NODES, OUT_SCALARS, OUT_ZONES, OUT_ALL = 1, 2, 4, 7

    def march(self, state, weather, zone_a0=None, zone_b0=None, outputs=None):
        """≙ ThermalModel::march: len(weather) sub-timesteps, in place on ``state``. ``outputs``: which of this
        path's outputs are written back (OUT_* bits; default all)."""
        w, n = as_weather(weather)
        a0, pa = self._opt(zone_a0)
        b0, pb = self._opt(zone_b0)
        if outputs is None:
            _check(self._L.heat_batch_march(self._h, self._state_ptr(state), state.size, w, n, pa, pb))
        else:
            _check(self._L.heat_batch_march_ex(self._h, self._state_ptr(state), state.size, w, n, pa, pb, int(outputs)))

    def download_outputs(self, state, outputs):
        _check(self._L.heat_batch_download_outputs(self._h, self._state_ptr(state), state.size, int(outputs)))

    def march_resident(self, weather, zone_a0=None, zone_b0=None):
        w, n = as_weather(weather)
        a0, pa = self._opt(zone_a0)
        b0, pb = self._opt(zone_b0)
        _check(self._L.heat_batch_march_resident(self._h, w, n, pa, pb))

    def synchronize(self):
        _check(self._L.heat_batch_synchronize(self._h))

    def failed_surface(self):
        """(index, kind) of the first place the last reported numerical failure was seen; (-1, 0) if none."""
        i, k = C.c_int64(-1), C.c_int32(0)
        _check(self._L.heat_batch_failed_surface(self._h, C.byref(i), C.byref(k)))
        return int(i.value), int(k.value)

    def set_weather(self, weather, zone_a0=None, zone_b0=None):
        w, n = as_weather(weather)
        a0, pa = self._opt(zone_a0)
        b0, pb = self._opt(zone_b0)
        _check(self._L.heat_batch_set_weather(self._h, w, n, pa, pb))

    def step_surfaces(self, sub_step):
        _check(self._L.heat_batch_step_surfaces(self._h, sub_step))

    def step_zones(self, gathered_ptr, n_blocks):
        _check(self._L.heat_batch_step_zones(self._h, gathered_ptr, n_blocks))

    def zone_partials_ptr(self):
        return self._L.heat_batch_zone_partials(self._h)

    def use_partials(self, dev_ptr):
        _check(self._L.heat_batch_use_partials(self._h, dev_ptr))

    def touched_zones(self):
        m = np.zeros(self.n_zones, dtype=np.uint8)
        _check(self._L.heat_batch_touched_zones(self._h, m.ctypes.data_as(C.POINTER(C.c_uint8))))
        return m

    def set_shared_zones(self, shared_zone):
        sz = np.ascontiguousarray(shared_zone, dtype=np.int32)
        _check(self._L.heat_batch_set_shared_zones(self._h, sz.ctypes.data_as(_i32p), len(sz)))

    def comm_init(self, unique_id, extra_shared=None):
        """ncclCommInitRank with the 128-byte id of comm_unique_id() (collective: every rank calls it), then the
        ranks agree on the shared zones (plus ``extra_shared``, tests). After it the batch marches like a
        single-GPU one."""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        if extra_shared is None or len(extra_shared) == 0:
            _check(self._L.heat_batch_comm_init(self._h, buf))
        else:
            ex = np.ascontiguousarray(extra_shared, dtype=np.int32)
            _check(self._L.heat_batch_comm_init_ex(self._h, buf, ex.ctypes.data_as(_i32p), len(ex)))

    def set_owned_zones(self, owned):
        m = np.ascontiguousarray(owned, dtype=np.uint8)
        assert len(m) == self.n_zones
        _check(self._L.heat_batch_set_owned_zones(self._h, m.ctypes.data_as(C.POINTER(C.c_uint8))))

    @property
    def n_shared_zones(self):
        return int(self._L.heat_batch_n_shared_zones(self._h))

    @property
    def comm_ranks(self):
        """Ranks of the batch's own RCCL communicator (0: none)."""
        return int(self._L.heat_batch_comm_ranks(self._h))

    def comm_destroy(self):
        _check(self._L.heat_batch_comm_destroy(self._h))

    def set_fusion(self, enabled):
        """Cluster-resident march on/off (off: every surface is streamed one sub-timestep per launch)."""
        _check(self._L.heat_batch_set_fusion(self._h, 1 if enabled else 0))

    @property
    def n_fused_surfaces(self):
        return int(self._L.heat_batch_n_fused_surfaces(self._h))

    @property
    def n_fused_launches(self):
        return int(self._L.heat_batch_n_fused_launches(self._h))

    def set_timing(self, enabled):
        _check(self._L.heat_batch_set_timing(self._h, int(enabled)))  # (k > 1: every k-th streamed march call)

    def get_timing(self):
        a, b, n = _d(0), _d(0), C.c_int64(0)
        _check(self._L.heat_batch_get_timing(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    @property
    def n_surfaces_in_batch(self):
        """Surfaces this batch holds (a shard of the model when created with rank_of_surface)."""
        return int(self._L.heat_batch_n_surfaces(self._h))

    @property
    def algorithmic_bytes(self):
        return self._L.heat_batch_algorithmic_bytes(self._h)

    @property
    def n_nodes(self):
        return self._L.heat_batch_n_nodes(self._h)

    def nomass_iterations(self):
        return self._L.heat_batch_nomass_iterations(self._h)

    def class_counts(self):
        c = (C.c_int64 * 5)()
        _check(self._L.heat_batch_class_counts(self._h, c))
        return list(c)


# ---------------------------------------------------------------------------------------------------
# Setup-time half (include/heat_amd_setup.h)
def make_layers(layers):
    """list of dicts -> ctypes array of heat_layer. Keys: thickness, and either is_gas/gas or k, rho, cp;
    optional front_thermal_abs, back_thermal_abs (0.84), tau (0), front_solar_abs, back_solar_abs (0.84) —
    the defaults the reference applies when the substance does not define the property."""
    arr = (Layer * len(layers))()
    for i, L in enumerate(layers):
        arr[i].is_gas = 1 if L.get("is_gas") else 0
        arr[i].gas = int(L.get("gas", 0))
        arr[i].thickness = float(L["thickness"])
        arr[i].conductivity = float(L.get("k", 0.0))
        arr[i].density = float(L.get("rho", 0.0))
        arr[i].specific_heat = float(L.get("cp", 0.0))
        arr[i].front_thermal_absorbtance = float(L.get("front_thermal_abs", 0.84))
        arr[i].back_thermal_absorbtance = float(L.get("back_thermal_abs", 0.84))
        arr[i].solar_transmittance = float(L.get("tau", 0.0))
        arr[i].front_solar_absorbtance = float(L.get("front_solar_abs", 0.84))
        arr[i].back_solar_absorbtance = float(L.get("back_solar_abs", 0.84))
    return arr


def discretize(layers, model_dt, max_dx, min_dt, height=1.0, angle=0.0):
    """Discretization::new (reference src/discretization.rs:95-114) through the C ABI."""
    L = load_library()
    arr = make_layers(layers)
    n_layers = len(layers)
    n_el = (C.c_int32 * n_layers)()
    sub = L.heat_discretize_construction(n_layers, arr, model_dt, max_dx, min_dt, n_el)
    if sub < 0:
        raise HeatError(sub, "heat_discretize_construction failed")
    return build_segments(layers, list(n_el), height, angle, tstep_subdivision=sub)


def build_segments(layers, n_elements, height=1.0, angle=0.0, tstep_subdivision=1):
    L = load_library()
    arr = make_layers(layers)
    n_layers = len(layers)
    n_el = (C.c_int32 * n_layers)(*n_elements)
    n_nodes = L.heat_count_nodes(n_layers, n_el)
    mass = np.zeros(n_nodes)
    uval = np.zeros(n_nodes)
    segc = np.zeros(n_nodes, dtype=np.int32)
    cav = np.zeros(max(n_layers, 1), dtype=CAVITY_DTYPE)
    nc = L.heat_build_segments(n_layers, arr, n_el, height, angle, mass.ctypes.data_as(_dp), uval.ctypes.data_as(_dp),
                               segc.ctypes.data_as(_i32p), cav.ctypes.data_as(C.POINTER(Cavity)), 0)
    if nc < 0:
        raise HeatError(nc, "heat_build_segments failed")
    fa = np.zeros(n_nodes)
    ba = np.zeros(n_nodes)
    rc = L.heat_node_alphas(n_layers, arr, n_el, n_nodes, fa.ctypes.data_as(_dp), ba.ctypes.data_as(_dp))
    return dict(tstep_subdivision=tstep_subdivision, n_elements=list(n_elements), n_nodes=n_nodes, mass=mass,
                uvalue=uval, seg_cavity=segc, cavities=cav[:nc].copy(), front_alpha=fa, back_alpha=ba, alpha_rc=rc)


def get_chunks(mass):
    L = load_library()
    mass = np.ascontiguousarray(mass, dtype=np.float64)
    n = len(mass)
    nm, nn = C.c_int32(0), C.c_int32(0)
    mc = (C.c_int32 * (2 * n + 2))()
    nc = (C.c_int32 * (2 * n + 2))()
    _check(L.heat_get_chunks(n, mass.ctypes.data_as(_dp), C.byref(nm), mc, C.byref(nn), nc))
    return ([(mc[2 * i], mc[2 * i + 1]) for i in range(nm.value)],
            [(nc[2 * i], nc[2 * i + 1]) for i in range(nn.value)])


class ModelBuilder:
    """ThermalModel::new (reference src/model.rs:215-354) through the C ABI: zones + surfaces with their
    constructions in, the flattened model dict (heat_amd.modeldict) + initial SimulationState out."""

    def __init__(self, n_per_hour, terrain=-1):
        self._L = load_library()
        self._h = self._L.heat_model_builder_create(n_per_hour, terrain)
        if not self._h:
            raise HeatError(-1, "heat_model_builder_create failed")
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._L.heat_model_builder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_zone(self, volume):
        return self._L.heat_model_builder_add_zone(self._h, float(volume))

    def add_surface(self, layers, area, perimeter, normal, centroid_z, front_kind, back_kind, front_zone=0,
                    back_zone=0, front_ambient=0.0, back_ambient=0.0, is_fenestration=False):
        arr = make_layers(layers)
        s = SurfaceIn()
        s.layers = arr
        s.n_layers = len(layers)
        s.is_fenestration = 1 if is_fenestration else 0
        s.area, s.perimeter, s.centroid_z = float(area), float(perimeter), float(centroid_z)
        for i in range(3):
            s.normal[i] = float(normal[i])
        s.front_kind, s.back_kind, s.front_zone, s.back_zone = int(front_kind), int(back_kind), int(front_zone), int(back_zone)
        s.front_ambient, s.back_ambient = float(front_ambient), float(back_ambient)
        rc = self._L.heat_model_builder_add_surface(self._h, C.byref(s))
        if rc < 0:
            raise HeatError(rc, "heat_model_builder_add_surface failed")
        return rc

    def finish(self):
        """Returns (model dict, initial state, dt_subdivisions)."""
        pd = C.POINTER(Desc)()
        ps = _dp()
        nsub = C.c_int32(0)
        rc = self._L.heat_model_builder_finish(self._h, C.byref(pd), C.byref(ps), C.byref(nsub))
        if rc != 0:
            raise HeatError(rc, "heat_model_builder_finish failed (the reference would panic or return Err here)")
        d = pd.contents
        S, Z, N = d.n_surfaces, d.n_zones, 0
        off = np.ctypeslib.as_array(d.node_offset, shape=(S + 1,)).copy()
        N = int(off[-1])

        def arr(p, n):
            return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0)

        md = dict(n_surfaces=int(S), n_zones=int(Z), n_state=int(d.n_state), dt=float(d.dt), node_offset=off,
                  front_hs_fix=None, back_hs_fix=None, seg_cavity=None, cavities=None)
        for k in ("mass", "uvalue", "front_alpha", "back_alpha"):
            md[k] = arr(getattr(d, k), N)
        for k in _I32:
            md[k] = arr(getattr(d, k), S).astype(np.int32)
        for k in [x for x in _F64 if x not in ("mass", "uvalue", "front_alpha", "back_alpha", "zone_volume")]:
            md[k] = arr(getattr(d, k), S)
        for k in [x for x in _I64 if x not in ("node_offset", "zone_slot")]:
            md[k] = arr(getattr(d, k), S).astype(np.int64)
        md["zone_volume"] = arr(d.zone_volume, Z)
        md["zone_slot"] = arr(d.zone_slot, Z).astype(np.int64)
        if d.n_cavities:
            md["seg_cavity"] = arr(d.seg_cavity, N).astype(np.int32)
            cv = np.zeros(d.n_cavities, dtype=CAVITY_DTYPE)
            C.memmove(cv.ctypes.data, d.cavities, cv.nbytes)
            md["cavities"] = cv
        state = np.ctypeslib.as_array(ps, shape=(int(d.n_state),)).copy() if int(d.n_state) > 0 else np.zeros(0)
        return md, state, int(nsub.value)

    def surface_info(self, i):
        sub, nn = C.c_int32(0), C.c_int32(0)
        el = (C.c_int32 * 64)()
        n = self._L.heat_model_builder_surface_info(self._h, i, C.byref(sub), C.byref(nn), el, 64)
        return dict(tstep_subdivision=sub.value, n_nodes=nn.value, n_elements=list(el[:n]))
