"""Builds heat_amd/lib/libheat_amd.so with hipcc for gfx950 (in-tree, so it travels with the repo)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libheat_amd.so")
SOURCES = ["kernels.hip", "batch.hip", "plan.cpp", "setup.cpp"]
HEADERS = ["layout.hpp", "kernels.hpp", "device_math.hpp", "plan.hpp", os.path.join("..", "..", "include", "heat_amd.h"),
           os.path.join("..", "..", "include", "heat_amd_setup.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


STAMPS_LIB = os.path.join(LIB_DIR, "libheat_amd_stamps.so")


def build_stamps(force=False):
    """Diagnostic build (-DHEAT_STAMPS): the cluster-resident workgroups stamp their phases (kernels.hip, HEAT_STAMP).
    Loaded through HEAT_AMD_LIB by tools/fused_phases.py only; not part of the product."""
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    if not force and os.path.exists(STAMPS_LIB) and os.path.getmtime(STAMPS_LIB) >= max(map(os.path.getmtime, deps)):
        return STAMPS_LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DHEAT_STAMPS", "-o", STAMPS_LIB] +
                          [os.path.join(CSRC, s) for s in SOURCES])
    return STAMPS_LIB


def build_variant(tag, defines, force=False):
    """An A/B build of the product library with extra -D switches: heat_amd/lib/libheat_amd_<tag>.so, loaded through
    HEAT_AMD_LIB by the measurement tools only (experiments; never the product)."""
    out = os.path.join(LIB_DIR, "libheat_amd_%s.so" % tag)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(map(os.path.getmtime, deps)):
        return out
    os.makedirs(LIB_DIR, exist_ok=True)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] +
                          ["-D" + d for d in defines] + ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES])
    return out


EXAMPLE_SRC = os.path.join(HERE, "..", "examples", "march_walls.cpp")
EXAMPLE_BIN = os.path.join(LIB_DIR, "march_walls")


def build_example(force=False):
    """examples/march_walls.cpp: a compiled (C++) host that drives the path through the C ABI only."""
    build()
    if not force and os.path.exists(EXAMPLE_BIN) and os.path.getmtime(EXAMPLE_BIN) >= max(
            os.path.getmtime(EXAMPLE_SRC), os.path.getmtime(LIB)):
        return EXAMPLE_BIN
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(HERE, "..", "include"), EXAMPLE_SRC,
                           "-L", LIB_DIR, "-lheat_amd", "-Wl,-rpath,$ORIGIN", "-o", EXAMPLE_BIN])
    return EXAMPLE_BIN


PLAN_HOST_LIB = os.path.join(LIB_DIR, "libheat_plan_host.so")


def build_plan_host(force=False):
    """The planner on its own (csrc/plan.cpp, host-only), compiled by g++ with AddressSanitizer + UBSan: the library
    tests/test_planner_host.py loads in a child process (LD_PRELOAD of libasan). Not part of the product."""
    src = os.path.join(CSRC, "plan.cpp")
    deps = [src, os.path.join(CSRC, "plan.hpp"), os.path.join(CSRC, "layout.hpp"),
            os.path.join(HERE, "..", "include", "heat_amd.h")]
    if not force and os.path.exists(PLAN_HOST_LIB) and os.path.getmtime(PLAN_HOST_LIB) >= max(map(os.path.getmtime, deps)):
        return PLAN_HOST_LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", src,
                           "-o", PLAN_HOST_LIB])
    return PLAN_HOST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_example(force="--force" in sys.argv))
