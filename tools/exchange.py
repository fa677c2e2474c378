"""Single-GPU rehearsal of the sharded sub-timestep (bench.py flags): fused march vs the sharded sequence with
0 / 16 zones declared shared, library-owned RCCL communicator vs torch.distributed. One rank: the all-gather
moves one block, so this measures the launch sequence, not xGMI."""
import json, subprocess, sys
base = [sys.executable, "bench.py", "--steps", "200", "--warmup", "20", "--no-cpu-baseline"]
for name, extra in (("fused", []), ("sharded native, nothing shared", ["--force-sharded"]),
                    ("16 shared zones, native", ["--force-shared-zones", "16"]),
                    ("16 shared zones, torch", ["--force-shared-zones", "16", "--collective", "torch"])):
    r = subprocess.run(base + extra, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-32s ms/step %.4f  value %.4g  kernel_us %.1f substep_us %.1f" % (
            name, j["ms_per_step"], j["value"], j["roofline"]["kernel_us"], j["roofline"]["substep_us"]), flush=True)
    except Exception as e:
        print(name, "FAILED", e, r.stderr[-3000:], flush=True)
