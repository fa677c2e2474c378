"""The C ABI under misuse (GPU): random sequences of calls on a small batch — marches of random lengths (also 0), resident
and on the caller's state, uploads / downloads of wrong sizes, sub-steps outside the weather set, NULL where a pointer is
wanted, timing switched on and off, fusion toggled, shared-zone sets that make sense and that do not — every call must
answer with 0 or an error code, and a batch that has been through it must still march right: the state is set again and a
last march compared with the oracle.      python tools/fuzz_api.py [SECONDS] [FIRST_SEED]"""
import ctypes as C, faulthandler, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from heat_amd import HeatBatch, HeatError, binding, modeldict as mdl
from oracle import oracle
from test_parity_gpu import assert_state_close

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle.lib()
L = binding.load_library()
t_end = time.time() + budget
n_cases = n_calls = n_errors = 0
while time.time() < t_end:
    faulthandler.dump_traceback_later(90, exit=True)
    rng = np.random.default_rng(seed)
    gen = [lambda: mdl.rooms_with_windows(int(rng.integers(40, 400)), Z=int(rng.integers(2, 9)), dt=45.0, seed=seed),
           lambda: mdl.clustered_massive(int(rng.integers(40, 400)), Z=int(rng.integers(2, 9)), dt=45.0, seed=seed),
           lambda: mdl.partitioned_buildings(int(rng.choice([96, 480, 960])), 9, rooms=int(rng.choice([4, 40])), dt=45.0, seed=seed)][seed % 3]
    md, st = gen()
    Z = int(md["n_zones"])
    kw = [dict(), dict(fuse_always=True), dict(no_fusion=True), dict(use_graph=True), dict(fuse_always=True, use_graph=True)][int(rng.integers(0, 5))]
    with HeatBatch(md, **kw) as b:
        h = b._h
        b.upload_state(st)
        host = st.copy()
        for _ in range(int(rng.integers(5, 40))):
            op = int(rng.integers(0, 14))
            n_calls += 1
            try:
                if op == 0:
                    b.march_resident(mdl.weather_series(int(rng.integers(0, 7)), 45.0))
                elif op == 1:
                    b.march(host, mdl.weather_series(int(rng.integers(0, 5)), 45.0), outputs=int(rng.integers(0, 9)))
                elif op == 2:  # a state of the wrong size
                    bad = np.zeros(max(1, len(st) + int(rng.choice([-1, 1, -len(st) + 1]))))
                    binding._check(L.heat_batch_upload_state(h, bad.ctypes.data_as(C.POINTER(C.c_double)), bad.size))
                elif op == 3:
                    binding._check(L.heat_batch_download_state(h, None, len(st)))
                elif op == 4:
                    b.set_weather(mdl.weather_series(int(rng.integers(0, 4)), 45.0))
                    b.step_surfaces(int(rng.integers(-1, 5)))
                elif op == 5:
                    b.set_timing(bool(rng.random() < 0.5))
                elif op == 6:
                    b.get_timing()
                elif op == 7:
                    b.set_fusion(bool(rng.random() < 0.5))
                elif op == 8:
                    b.synchronize()
                elif op == 9:
                    b.download_outputs(host, int(rng.integers(0, 9)))
                elif op == 10:
                    b.upload_inputs(host)
                elif op == 11:  # zone lists that make no sense
                    b.set_shared_zones(np.array([int(rng.integers(-2, Z + 3))], dtype=np.int32))
                elif op == 12:
                    b.failed_surface(); b.nomass_iterations(); b.class_counts()
                else:
                    binding._check(L.heat_batch_march_resident(h, None, int(rng.integers(-1, 3)), None, None))
            except HeatError:
                n_errors += 1
            except AssertionError:
                n_errors += 1
        # after all that the batch must still be what it was: the state again, nothing shared, fusion as planned, one march
        try:
            b.set_shared_zones(np.zeros(0, dtype=np.int32))
        except HeatError:
            pass
        b.set_fusion(True)
        b.set_timing(False)
        b.upload_state(st)
        w = mdl.weather_series(int(rng.integers(1, 6)), 45.0, wind_speed=2.0, wind_deg=100.0)
        ref = st.copy()
        rc, iters = oracle.OracleModel(md).march(ref, w)
        got = st.copy()
        b.march(got, w)
        assert_state_close(md, ref, got)
    n_cases += 1
    seed += 1
print("fuzz_api: %d batches, %d calls (%d answered with an error code), every batch marched right afterwards; seeds up to %d" % (n_cases, n_calls, n_errors, seed - 1))
