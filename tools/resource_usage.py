"""Per-kernel register / scratch / LDS / occupancy table of a -Rpass-analysis=kernel-resource-usage log:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c heat_amd/csrc/kernels.hip -o /tmp/k.o -Rpass-analysis=kernel-resource-usage 2> ru.txt
    python tools/resource_usage.py ru.txt"""
import re, subprocess, sys
rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: +(\w[\w ]*?)(?: \[[\w/]+\])?: +(\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k in ("Function Name", "Name"):
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("heat::", "").replace("void ", "")
    print("%-46s VGPR %4s AGPR %3s SGPR %4s scratch %5s LDS %6s occupancy %s" % (
        n, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs") or r.get("SGPRs"), r.get("ScratchSize"),
        r.get("LDS Size"), r.get("Occupancy")))
