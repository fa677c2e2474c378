import sys, os, time, json, subprocess
# runs bench.py with several HEAT_AMD_VARIANT values (each a fresh process) and prints kernel_us
for var in sys.argv[1:]:
    env = dict(os.environ, HEAT_AMD_VARIANT=var)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "100", "--warmup", "10", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    try:
        r = json.loads(out.stdout.strip().splitlines()[-1])
        print("VARIANT", var, "kernel_us %.1f" % r["roofline"]["kernel_us"], "substep_us %.1f" % r["roofline"]["substep_us"], "frac %.3f" % r["roofline"]["frac"], flush=True)
    except Exception as e:
        print("VARIANT", var, "failed", out.stdout[-300:], out.stderr[-500:])
