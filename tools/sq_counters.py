"""Per-kernel means of the counters collected by tools/pmc_passes.sh:
    python tools/sq_counters.py OUT_DIR [kernel substring] [sub-timesteps per launch] > profiles/<tag>_sq_counters.json"""
import csv, glob, json, os, sys
d = sys.argv[1]
kname = sys.argv[2] if len(sys.argv) > 2 else ""
sub = int(sys.argv[3]) if len(sys.argv) > 3 else 1
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kname not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"][:60]
        a = acc.setdefault(k, {})
        c = a.setdefault(r["Counter_Name"], [0.0, 0, 0.0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
        if "End_Timestamp" in r and r["End_Timestamp"]:
            c[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {}
for k, a in acc.items():
    m = {c: v[0] / v[1] for c, v in a.items()}
    o = {"launches_sampled": max(v[1] for v in a.values()), "sub_timesteps_per_launch": sub}
    o["mean_duration_us_under_pmc"] = {c: v[2] / v[1] for c, v in a.items() if v[2] > 0}
    o["counters_per_launch"] = m
    w = m.get("SQ_WAVES")
    if w:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if c in m:
                o[c.replace("SQ_INSTS_", "") + "_per_wave_per_sub_timestep"] = m[c] / w / sub
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in m:
                o[c + "_frac_of_wave_cycles"] = m[c] / wc
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        o["hbm_traffic_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * 2 + m["WRITE_SIZE"] * 1024
        o["hbm_traffic_note"] = "FETCH_SIZE [KB] x 1024 x 2 (gfx950: 64 B booked per 128-B request of 16-B-per-lane loads) + WRITE_SIZE [KB] x 1024"
    out[k] = o
print(json.dumps(out, indent=1))
