"""Child process of tests/test_planner_host.py: runs the planner (heat_amd/csrc/plan.cpp, built by g++ with
AddressSanitizer + UBSan) over many models. Started with LD_PRELOAD=libasan; any sanitizer report aborts it."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heat_amd import binding, modeldict as mdl  # noqa: E402
from tests.helpers import random_zone_graph_model, walls_example_model  # noqa: E402


def main(path):
    L = binding.load_host_library(path)
    n_plans = 0
    # the random zone graphs of test_planner_stress_random_zone_graphs, all three fusion modes and blocking factors
    for seed in range(40):
        md, _ = random_zone_graph_model(seed)
        for opts in (dict(), dict(no_fusion=True), dict(fuse_always=True), dict(fuse_always=True, nodes_per_lane=4),
                     dict(fuse_always=True, nodes_per_lane=8), dict(fuse_always=True, nodes_per_lane=16),
                     dict(no_palette=True), dict(force_general=True)):
            binding.plan_check(md, lib=L, **opts)
            n_plans += 1
        for n_ranks in (1, 2, 3, 8):
            ranks, n_shared = binding.partition(md, n_ranks, lib=L)
            assert ranks.min() >= 0 and ranks.max() < n_ranks
    # the 12-wall, 2-zone model of examples/march_walls.cpp
    md, _ = walls_example_model()
    for opts in (dict(), dict(fuse_always=True), dict(no_fusion=True)):
        binding.plan_check(md, lib=L, **opts)
        n_plans += 1
    # the synthetic workloads at test sizes
    for gen, kw in ((mdl.ragged_mixed, dict(S=3000, Z=30)), (mdl.clustered_massive, dict(S=2000, Z=80)),
                    (mdl.rooms_with_windows, dict(S=2000, Z=100)), (mdl.glazing_cavity, dict(S=400, Z=4)),
                    (mdl.uniform_massive, dict(S=3000, n=32, Z=30)), (mdl.uniform_massive, dict(S=1, n=13, Z=1))):
        md, _ = gen(**kw)
        for opts in (dict(), dict(fuse_always=True), dict(no_fusion=True)):
            binding.plan_check(md, lib=L, **opts)
            n_plans += 1
    # walls of many materials (wide palettes, layout.hpp; beyond them per-node constants), with and without facings
    for seed, layers in ((3, 5), (4, 6), (5, 9)):
        md, _ = random_zone_graph_model(seed)
        rng = np.random.default_rng(seed)
        off = md["node_offset"]
        mass, u = md["mass"].copy(), md["uvalue"].copy()
        for s_ in range(0, md["n_surfaces"], 2):
            n = int(off[s_ + 1] - off[s_])
            for layer in range(layers):
                a, b_ = off[s_] + n * layer // layers, off[s_] + n * (layer + 1) // layers
                mass[a:b_] *= rng.uniform(0.7, 1.3)
                u[a:b_] *= rng.uniform(0.7, 1.3)
        md["mass"], md["uvalue"] = mass, u
        for opts in (dict(), dict(no_fusion=True), dict(fuse_always=True), dict(fuse_always=True, nodes_per_lane=16)):
            binding.plan_check(md, lib=L, **opts)
            n_plans += 1
    # degenerate descriptors
    md, _ = mdl.uniform_massive(0, 8, Z=0)
    binding.plan_check(md, lib=L)
    md, _ = mdl.uniform_massive(5, 8, Z=3)
    md["back_zone"] = np.array([0, 0, 0, 0, 7], dtype=np.int32)  # out of range: must be refused, not read
    try:
        binding.plan_check(md, lib=L)
    except binding.HeatError as e:
        assert e.code == -4, e
    else:
        raise AssertionError("zone out of range was accepted")
    print("planner host check: %d plans verified" % n_plans)


if __name__ == "__main__":
    main(sys.argv[1])
