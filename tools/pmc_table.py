"""Per-kernel means of every counter in the rocprofv3 --pmc output directories given (one or more passes):
    python tools/pmc_table.py DIR [DIR ...] [--kernel SUBSTR]"""
import csv, glob, os, sys, collections
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
sub = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "k_surfaces"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0, 0.0]))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void heat::", "")
            if sub not in k:
                continue
            c = acc[k][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"]); c[1] += 1
            c[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, cs in acc.items():
    print(k)
    m = {c: v[0] / v[1] for c, v in cs.items()}
    for c in sorted(cs):
        v = cs[c]
        print("  %-28s %16.0f   (%d dispatches, %.1f us under pmc)" % (c, v[0] / v[1], v[1], v[2] / v[1]))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
            if c in m:
                print("  %-28s %.3f of wave-cycles" % (c, m[c] / wc))
    if "SQ_INSTS_VALU" in m and m.get("SQ_WAVES"):
        print("  VALU per wave %.0f" % (m["SQ_INSTS_VALU"] / m["SQ_WAVES"]))
    f64 = [m.get("SQ_INSTS_VALU_%s_F64" % x) for x in ("ADD", "MUL", "FMA", "TRANS")]
    if all(v is not None for v in f64):
        print("  f64 insts %.0f (add %.0f mul %.0f fma %.0f trans %.0f)" % (sum(f64), *f64))
