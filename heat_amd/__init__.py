"""heat_amd — MI355X-native wall heat-conduction path (drop-in for the hot path of
SIMPLE-BuildingSimulation/heat: ThermalModel::march / iterate_surfaces / rk4).

The product is the C-ABI library ``heat_amd/lib/libheat_amd.so`` (include/heat_amd.h),
built from the HIP sources under ``heat_amd/csrc``. This Python package is plumbing:
a ctypes binding (``HeatBatch``), the state-slot layout helper and the synthetic
workloads used by tests and bench. There is no CPU fallback: creating a batch without the
HIP library or without a GPU raises.
"""
from .binding import HeatBatch, HeatError, ModelBuilder, Weather, lib_path, load_library, build_library  # noqa: F401
from . import modeldict  # noqa: F401

__all__ = ["HeatBatch", "HeatError", "ModelBuilder", "Weather", "lib_path", "load_library", "build_library", "modeldict"]
