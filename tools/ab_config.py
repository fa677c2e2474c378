"""A/B of library builds on one bench workload, interleaved on one box:
    python tools/ab_config.py CONFIG MODE LIB [LIB ...]     (LIB: path, or 'default')"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg, mode, libs = sys.argv[1], sys.argv[2], sys.argv[3:]
for rep in range(3):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["HEAT_AMD_LIB"] = os.path.join(ROOT, "heat_amd", "lib", "libheat_amd_%s.so" % lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_config.py"), cfg, "20", "3", mode], env=env,
                             capture_output=True, text=True).stdout.strip()
        print("%-8s %s" % (lib, out[out.find("| wall"):]), flush=True)
