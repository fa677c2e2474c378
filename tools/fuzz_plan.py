"""The fuzzer's cases (tools/fuzz.py, make_case) through the host-only planner built with -fsanitize=address,undefined:
heat_plan_check re-derives every invariant of the plan, the sanitizers watch the planner's memory. No GPU.
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/fuzz_plan.py [SECONDS] [FIRST_SEED]
Every fourth case is also cut into shards (heat_partition or arbitrary ranges) and every shard planned."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
stub = types.ModuleType("test_parity_gpu")      # (fuzz.py imports the GPU tests' comparison; not needed here)
stub.assert_state_close = lambda *a: None
sys.modules["test_parity_gpu"] = stub
import numpy as np
import fuzz
from heat_amd import binding, build as hb

L = binding.load_host_library(hb.build_plan_host())
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
t_end = time.time() + budget
n = n_shards = 0
while time.time() < t_end:
    md, st, name, w, a0, b0, kw, cuts = fuzz.make_case(seed)
    kw = {k: v for k, v in kw.items() if k != "use_graph"}
    try:
        binding.plan_check(md, lib=L, **kw)
        if seed % 4 == 3 and int(md["n_surfaces"]) >= 16:
            rng = np.random.default_rng(seed ^ 0x5eed)
            S = int(md["n_surfaces"])
            R = int(rng.choice([2, 3, 4, 8]))
            if rng.random() < 0.5:
                ranks, _ = binding.partition(md, R, lib=L)
            else:
                edges = np.sort(rng.choice(np.arange(1, S), R - 1, replace=False))
                ranks = np.searchsorted(edges, np.arange(S), side="right").astype(np.int32)
            from heat_amd.sharded import shard_by_ranks
            for r in range(R):
                if (ranks == r).any():
                    binding.plan_check(shard_by_ranks(md, ranks, r), lib=L, n_ranks=R, rank=r, **kw)
                    n_shards += 1
        n += 1
    except Exception as e:  # noqa
        print("FAIL seed %d %s %s: %s" % (seed, name, kw, str(e)[:400]), flush=True)
    seed += 1
print("fuzz_plan: %d models, %d shards planned and checked, seeds up to %d" % (n, n_shards, seed - 1))
