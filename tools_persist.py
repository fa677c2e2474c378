import sys, os, subprocess, json
for tune in sys.argv[1:]:
    env = dict(os.environ, HEAT_AMD_PERSIST=tune)
    for npl in ("8", "16"):
        out = subprocess.run([sys.executable, "bench.py", "--steps", "100", "--warmup", "10", "--no-cpu-baseline", "--nodes-per-lane", npl], env=env, capture_output=True, text=True)
        try:
            r = json.loads(out.stdout.strip().splitlines()[-1])
            print("PERSIST", tune, "npl", npl, "kernel_us %.1f" % r["roofline"]["kernel_us"], "substep_us %.1f" % r["roofline"]["substep_us"], "frac %.3f" % r["roofline"]["frac"], flush=True)
        except Exception as e:
            print("PERSIST", tune, "failed", out.stdout[-300:], out.stderr[-800:])
