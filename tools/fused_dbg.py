import os, subprocess, sys
for dbg in (0, 8, 1, 3, 11):
    env = dict(os.environ, HEAT_AMD_FUSED_DEBUG=str(dbg))
    r = subprocess.run([sys.executable, "tools/fused_time.py", "1000000", "32", "20"], capture_output=True, text=True, env=env)
    print("debug bits", dbg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
