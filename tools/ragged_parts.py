"""Where BASELINE config 3's streamed sub-timestep goes, by the make-up of its walls: the all-massive walls, the walls
with no-mass facings and the two-node no-mass walls of the same model, each streamed on its own (and by node count)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from heat_amd import HeatBatch, modeldict as mdl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
md, st = mdl.ragged_mixed(S, Z=max(1, S // 100), dt=45.0, seed=20260401)
off = np.asarray(md["node_offset"]); n = np.diff(off)
m0 = np.asarray(md["mass"])[off[:-1]]
kinds = {"all": np.arange(S), "massive": np.nonzero(m0 >= 1e-5)[0], "faced": np.nonzero((m0 < 1e-5) & (n > 2))[0],
         "two-node": np.nonzero(n == 2)[0]}
kinds["massive n<=32"] = kinds["massive"][n[kinds["massive"]] <= 32]
kinds["massive n>32"] = kinds["massive"][n[kinds["massive"]] > 32]
kinds["faced n<=32"] = kinds["faced"][n[kinds["faced"]] <= 32]
kinds["faced n>32"] = kinds["faced"][n[kinds["faced"]] > 32]
w = mdl.weather_series(20, 45.0)
for name, idx in kinds.items():
    sub = mdl.subset(md, idx)
    with HeatBatch(sub, use_graph=True, no_fusion=True) as b:
        b.upload_state(st)
        b.march_resident(w); b.synchronize()
        b.set_timing(True)
        b.march_resident(w); b.synchronize()
        su, ss, _ = b.get_timing()
        nodes = int(sub["node_offset"][-1])
        print("%-14s %8d walls %9d nodes classes %s: surfaces %.1f us (%.2f ps/node), sub-timestep %.1f us, passes/sub-timestep %d" % (
            name, len(idx), nodes, b.class_counts(), su, su * 1e6 / max(nodes, 1), ss, b.nomass_iterations() // 41), flush=True)
