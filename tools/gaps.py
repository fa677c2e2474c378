"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv):
    python tools/gaps.py TRACE.csv [min_kernel_us]
Prints per kernel name the mean gap in front of it, and the total busy / idle time of the trace's main span."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps = collections.defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None:
        gaps[r["Kernel_Name"][:60]].append((s - prev_end) / 1e3)
    prev_end = max(prev_end or 0, e)
for k, v in gaps.items():
    v2 = sorted(v)
    print("%-62s n=%4d  gap before it: median %8.1f us  mean %8.1f us" % (k, len(v), v2[len(v2) // 2], sum(v) / len(v)))
