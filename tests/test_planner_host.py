"""The planner (heat_amd/csrc/plan.cpp: classification, zone-connected clusters, workgroups of the cluster-resident
march, tiles, packing, partition over ranks) is host-only code. Here it is compiled by g++ with AddressSanitizer and
UBSan and run, in a child process, over the random zone graphs of the GPU planner stress test, the 12-wall model of
examples/march_walls.cpp and the synthetic workloads — every plan is verified by heat_plan_check (every surface in
exactly one tile, every index inside its array, every workgroup inside the kernel's limits). No GPU needed.

Reference: the planner decides how `iterate_surfaces` (src/model.rs:102-180) and the zone balance
(src/model.rs:556-590) are laid out on the device; it has no counterpart in the reference."""
import os
import subprocess
import sys

import numpy as np
import pytest

from heat_amd import binding, build as hb, modeldict as mdl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_planner_under_address_and_ub_sanitizers():
    asan = _asan_runtime()
    if asan is None:
        pytest.skip("gcc has no libasan here")
    lib = hb.build_plan_host()
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "planner_host_worker.py"), lib], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-6000:])
    assert "plans verified" in out.stdout


def test_fuzzer_cases_through_the_sanitized_planner():
    """tools/fuzz_plan.py: the GPU fuzzer's random cases (every generator, random rewiring of the zone graph, planner modes,
    blocking factors, shards by heat_partition or by arbitrary ranges) through the host-only planner under the sanitizers,
    every plan re-derived by heat_plan_check — a short run of it here (2 800 models clean in four minutes in round 3)."""
    asan = _asan_runtime()
    if asan is None:
        pytest.skip("gcc has no libasan here")
    hb.build_plan_host()
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_plan.py"), "12", "300000"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-6000:])
    assert "FAIL" not in out.stdout and "models," in out.stdout, out.stdout[-2000:]


def test_damaged_descriptors_are_refused_or_planned_never_crash():
    """tools/fuzz_desc.py: a valid small model with one to three fields damaged (indices out of range, negative slots, NaN /
    infinite / negative physical values, node offsets out of order, dt <= 0 ...) through the sanitized planner: an error code
    or a plan — a short run of it (82 000 clean in round 3, after the NaN thermal mass it found: an endless loop)."""
    asan = _asan_runtime()
    if asan is None:
        pytest.skip("gcc has no libasan here")
    hb.build_plan_host()
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_desc.py"), "6", "5000"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "no crash" in out.stdout, (out.stdout[-2000:], out.stderr[-6000:])


def test_nan_thermal_mass_is_refused(host_lib):
    md, _ = mdl.uniform_massive(20, 9, Z=2, dt=45.0)
    m = np.array(md["mass"], dtype=np.float64)
    m[17] = np.nan
    md["mass"] = m
    with pytest.raises(binding.HeatError) as e:
        binding.plan_check(md, lib=host_lib)
    assert "thermal mass is NaN" in str(e.value)


@pytest.fixture(scope="module")
def host_lib():
    # the product library exports the same host-only entry points; they need no device
    return binding.load_library()


def test_partition_keeps_clusters_whole(host_lib):
    """Isolated zones (the headline's topology): shards cut along zone boundaries share nothing."""
    md, _ = mdl.uniform_massive(12000, 8, Z=120, dt=45.0)
    for n_ranks in (2, 3, 8):
        ranks, n_shared = binding.partition(md, n_ranks, lib=host_lib)
        assert n_shared == 0
        # every zone's walls sit on one rank
        for z in (0, 17, 119):
            assert len(np.unique(ranks[md["back_zone"] == z])) == 1
        # balanced to a cluster's weight
        w = np.bincount(ranks, minlength=n_ranks)
        assert w.max() - w.min() <= 2 * 100, w
        # model order is kept: rank numbers never decrease along the surfaces
        assert (np.diff(ranks) >= 0).all()


def test_partition_cuts_only_oversized_clusters(host_lib):
    """BASELINE config 3's zones form one ring (every Space/Space wall joins neighbours): one cluster holds the whole
    model and has to be cut; zones are shared at the cuts only."""
    md, _ = mdl.ragged_mixed(20000, Z=200, dt=45.0)
    ranks, n_shared = binding.partition(md, 4, lib=host_lib)
    assert 1 <= n_shared <= 3 * 4, n_shared
    nodes = np.diff(md["node_offset"])
    w = np.bincount(ranks, weights=32.0 * nodes + 152.0, minlength=4)
    assert w.max() / w.min() < 1.02, w
    # small clusters beside one oversized one: only the big one is cut
    md, _ = mdl.clustered_massive(6000, Z=240, dt=45.0)
    big = np.arange(6000) < 3000
    md["front_kind"] = np.where(big, mdl.SPACE, md["front_kind"]).astype(np.int32)
    md["front_zone"] = np.where(big, (md["back_zone"] + 1) % 120, md["front_zone"]).astype(np.int32)
    md["back_zone"] = np.where(big, md["back_zone"] % 120, md["back_zone"]).astype(np.int32)
    ranks, n_shared = binding.partition(md, 3, lib=host_lib)
    touched = {}
    for s in range(6000):
        for kind, zone in ((md["front_kind"][s], md["front_zone"][s]), (md["back_kind"][s], md["back_zone"][s])):
            if kind == mdl.SPACE:
                touched.setdefault(int(zone), set()).add(int(ranks[s]))
    shared = [z for z, r in touched.items() if len(r) > 1]
    assert len(shared) == n_shared
    assert all(z < 120 for z in shared), shared     # only zones of the ring are shared


def test_plan_check_reports_the_plan(host_lib):
    md, _ = mdl.clustered_massive(1500, Z=60, dt=45.0)
    s = binding.plan_check(md, lib=host_lib, fuse_always=True)
    assert sum(s[:5]) == 1500 and s[5] > 0 and s[6] > 0 and s[7] > 0
    s = binding.plan_check(md, lib=host_lib, no_fusion=True)
    assert sum(s[:5]) == 1500 and s[5] == 0 and s[6] == 0
