"""profiles/<tag>_counters.json from the passes of tools/pmc_passes.sh — the file bench.py reads for `roofline`:
    python tools/make_counters.py TAG OUT_DIR KERNEL_SUBSTRING CONFIG SURFACES NODES_TOTAL MODE SUBSTEPS_PER_LAUNCH
MODE: streamed (one sub-timestep per launch) or fused (cluster-resident march)."""
import csv, glob, hashlib, json, os, sys
tag, d, kname, config, S, N, mode, sub = sys.argv[1:9]
S, N, sub = int(S), int(N), int(sub)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (a sub-timestep may be several kernels — the streamed parts of k_surfaces_stream<V> —: per counter the means of every
# matching kernel are ADDED; every such kernel runs once per sub-timestep)
per_kernel = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kname not in r["Kernel_Name"]:
            continue
        kn = r["Kernel_Name"].split("(")[0]
        c = per_kernel.setdefault(kn, {}).setdefault(r["Counter_Name"], [0.0, 0, 0.0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
        c[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
acc = {}
for kn, cs in per_kernel.items():
    for cn, v in cs.items():
        a = acc.setdefault(cn, [0.0, 0, 0.0])
        a[0] += v[0] / v[1]; a[1] = max(a[1], v[1]); a[2] += v[2] / v[1]
m = {c: v[0] for c, v in acc.items()}
# the kernel sources these counters were measured on: bench.py prints "counters_stale": true (and drops every figure
# derived from this file) when its own hash of the same files differs
sys.path.insert(0, ROOT)
import bench
j = {"kernel_sources_sha256": bench.kernel_source_hash(), "kernel_sources": list(bench.KERNEL_SOURCES),
     "workload": {"config": config, "surfaces": S, "nodes_total": N, "mode": mode, "substeps_per_launch": sub},
     "kernel": kname, "kernels_matched": sorted(per_kernel), "dispatches_sampled": {c: v[1] for c, v in acc.items()},
     "mean_duration_us_under_pmc": {c: v[2] for c, v in acc.items()},
     "counters_per_launch": m}
f64 = [m.get("SQ_INSTS_VALU_%s_F64" % k) for k in ("ADD", "MUL", "FMA", "TRANS")]
if all(v is not None for v in f64):
    # wave-instructions by class: adds and multiplies 1 flop per lane, FMAs 2; 64 lanes
    j["f64_insts_per_launch"] = sum(f64)
    j["f64_flops_per_launch"] = (f64[0] + f64[1] + 2 * f64[2]) * 64
if "SQ_INSTS_VALU" in m:
    j["valu_insts_per_launch"] = m["SQ_INSTS_VALU"]
    j["waves_per_launch"] = m.get("SQ_WAVES")
    if m.get("SQ_WAVES"):
        j["valu_insts_per_wave_per_sub_timestep"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"] / sub
wc = m.get("SQ_WAVE_CYCLES")
if wc:
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        if c in m:
            j[c + "_frac_of_wave_cycles"] = m[c] / wc
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    j["hbm_traffic_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * 2 + m["WRITE_SIZE"] * 1024
    j["hbm_traffic_note"] = ("FETCH_SIZE [KB] x 1024 x 2 (gfx950: the counter books 64 B per 128-B request of 16-B-per-lane "
                             "loads, MI355X_MICROARCH.md) + WRITE_SIZE [KB] x 1024; separate --pmc passes")
out = os.path.join(ROOT, "profiles", "%s_counters.json" % tag)
json.dump(j, open(out, "w"), indent=1)
print(out)
print(json.dumps({k: v for k, v in j.items() if k != "counters_per_launch"}, indent=1))
