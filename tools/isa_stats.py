"""Instruction mix and spill sites of one kernel in a hipcc -S listing:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only heat_amd/csrc/kernels.hip -o /tmp/k.s
    python tools/isa_stats.py /tmp/k.s k_surfaces_stream"""
import collections, re, sys
txt = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and name in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l)
end = next(i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end"))
ins = [l.strip() for l in txt[start + 1:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
print("instructions:", len(ins))
c = collections.Counter(i.split()[0] for i in ins)
groups = collections.Counter()
for op, n in c.items():
    g = ("scratch" if op.startswith("scratch") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else
         "lds" if op.startswith("ds_") else "salu/smem" if op.startswith("s_") else
         "valu f64" if op.startswith("v_") and "f64" in op else "valu other" if op.startswith("v_") else "other")
    groups[g] += n
print(dict(groups))
print(c.most_common(30))
sites = [n for n, i in enumerate(ins) if i.startswith("scratch_")]
print("scratch instructions:", len(sites), "at", sites[:80])
