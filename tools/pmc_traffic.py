"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into
profiles/<tag>_pmc_traffic.json and per-dispatch CSVs.

    python tools/pmc_traffic.py <tag> <fetch_dir> <write_dir> <kernel substring> <surfaces> <nodes> <mode> <substeps_per_launch>

HBM bytes per launch = FETCH_SIZE [KB] x 1024 x 2 (gfx950: the counter books 64 B per 128-B request for
16-byte-per-lane streaming loads, MI355X_MICROARCH.md) + WRITE_SIZE [KB] x 1024."""
import csv, glob, json, os, sys
tag, fdir, wdir, kname, S, n, mode, sub = sys.argv[1:9]
S, n, sub = int(S), int(n), int(sub)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and kname in r["Kernel_Name"]]
    out = os.path.join(ROOT, "profiles", "%s_pmc_%s.csv" % (tag, counter))
    with open(out, "w") as fo:
        fo.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value,Duration_us\n")
        for r in rows:
            fo.write('%s,"%s",%s,%s,%.1f\n' % (r["Dispatch_Id"], r["Kernel_Name"][:60], counter, r["Counter_Value"],
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    vals = [float(r["Counter_Value"]) for r in rows]
    return sum(vals) / len(vals), len(vals)


fetch, nf = collect(fdir, "FETCH_SIZE")
write, nw = collect(wdir, "WRITE_SIZE")
j = {"workload": {"surfaces": S, "nodes": n, "mode": mode, "substeps_per_launch": sub},
     "kernel": kname, "dispatches_sampled": [nf, nw],
     "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
     "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane streaming loads -> x2 "
                   "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
     "read_bytes": fetch * 1024 * 2, "write_bytes": write * 1024,
     "traffic_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
     "algorithmic_bytes_per_launch": (32 * S * n + 152 * S) * sub}
json.dump(j, open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag), "w"), indent=1)
print(json.dumps(j, indent=1))
