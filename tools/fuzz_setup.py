"""The product's setup-time library (heat_amd/csrc/setup.cpp: discretize_construction, build, get_chunks, the alpha
distribution) against the oracle's independent restatement on random constructions, value for value. CPU only.
    python tools/fuzz_setup.py [SECONDS] [FIRST_SEED]"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from heat_amd import binding, modeldict as mdl
from oracle import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle.lib()
t_end = time.time() + budget
n = bad = rejected = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    nl = int(rng.integers(1, 7))
    layers = []
    for i in range(nl):
        gas_ok = 0 < i < nl - 1 and not layers[-1].get("is_gas")
        if gas_ok and rng.random() < 0.25:
            layers.append(dict(thickness=float(rng.uniform(0.004, 0.08)), is_gas=True, gas=int(rng.choice([mdl.AIR, mdl.ARGON]))))
            continue
        L = dict(thickness=float(10 ** rng.uniform(-3, -0.3)), k=float(10 ** rng.uniform(-1.7, 0.5)),
                 rho=float(10 ** rng.uniform(1, 3.5)), cp=float(rng.uniform(400., 2600.)))
        if rng.random() < 0.5:
            L.update(front_thermal_abs=float(rng.uniform(0, 1)), back_thermal_abs=float(rng.uniform(0, 1)),
                     front_solar_abs=float(rng.uniform(0, 1)), back_solar_abs=float(rng.uniform(0, 1)))
        if rng.random() < 0.1:
            L.update(tau=float(rng.uniform(0.1, 0.9)), front_solar_abs=float(rng.uniform(0, 0.1)), back_solar_abs=float(rng.uniform(0, 0.1)))
        layers.append(L)
    main_dt = float(rng.choice([60., 90., 180., 300., 600., 900., 1800., 3600.]))
    max_dx = float(rng.choice([0.04, 0.02, 0.1]))
    min_dt = float(rng.choice([60., 30., 120.]))
    angle = float(rng.uniform(0, math.pi))
    ra = rb = None
    try:
        a = binding.discretize(layers, main_dt, max_dx, min_dt, 1., angle)
    except Exception as e:  # noqa
        ra = str(e)[:80]
    try:
        b = oracle.discretize(layers, main_dt, max_dx, min_dt, 1., angle)
    except Exception as e:  # noqa
        rb = str(e)[:80]
    if ra or rb:
        if bool(ra) != bool(rb):
            bad += 1
            print("FAIL seed %d: one side rejects: product %r oracle %r" % (seed, ra, rb), flush=True)
        else:
            rejected += 1
        seed += 1
        continue
    # (a mixture of transparent and opaque layers: the reference panics, surface.rs:470-472,506-508 — both sides say so with
    # a negative alpha_rc, the code itself is theirs; the alphas are then unspecified)
    both_reject = a["alpha_rc"] < 0 and b["alpha_rc"] < 0
    if both_reject:
        rejected += 1
    ok = a["tstep_subdivision"] == b["tstep_subdivision"] and a["n_elements"] == b["n_elements"] and (both_reject or a["alpha_rc"] == b["alpha_rc"])
    for k in ("mass", "uvalue") + (() if both_reject else ("front_alpha", "back_alpha")):
        ok = ok and np.array_equal(a[k], b[k], equal_nan=True)
    ok = ok and np.array_equal(a["seg_cavity"], b["seg_cavity"])
    for f in ("thickness", "height", "angle", "eout", "ein", "gas"):
        ok = ok and np.array_equal(a["cavities"][f], b["cavities"][f])
    if len(b["mass"]) <= oracle.MAX_NODES:  # (the oracle's scratch arrays hold 1 024 nodes per surface)
        ok = ok and binding.get_chunks(a["mass"]) == oracle.get_chunks(b["mass"])
    if not ok:
        bad += 1
        print("FAIL seed %d: %d layers main_dt %g: product and oracle differ (n_elements %s / %s, subdivision %s / %s)" % (
            seed, nl, main_dt, a["n_elements"], b["n_elements"], a["tstep_subdivision"], b["tstep_subdivision"]), flush=True)
    n += 1
    seed += 1
print("fuzz_setup: %d constructions equal, %d rejected by both, %d differ" % (n, rejected, bad))

# damaged constructions through the product's library alone: an error code or a result, never a crash or an endless loop
# (a non-positive thickness or conductivity, NaN / infinite properties, a gas gap at a face, absurd timesteps)
import signal
def _alarm(*a):
    raise TimeoutError("the setup library did not come back")
signal.signal(signal.SIGALRM, _alarm)
rng = np.random.default_rng(seed)
n_dmg = n_refused = 0
for _ in range(4000):
    nl = int(rng.integers(1, 5))
    layers = []
    for i in range(nl):
        if rng.random() < 0.2:
            layers.append(dict(thickness=float(rng.choice([0.01, 0.0, -0.01, np.nan, 1e9])), is_gas=True, gas=int(rng.choice([mdl.AIR, mdl.ARGON, 7, -1]))))
            continue
        L = dict(thickness=float(rng.choice([0.1, 0.0, -1.0, np.nan, np.inf, 1e-12, 1e6])), k=float(rng.choice([1.0, 0.0, -1.0, np.nan, np.inf, 1e-300])),
                 rho=float(rng.choice([1000., 0.0, -5.0, np.nan, 1e300])), cp=float(rng.choice([1000., 0.0, -1.0, np.nan, np.inf])))
        if rng.random() < 0.3:
            L.update(tau=float(rng.choice([0.5, -1.0, 2.0, np.nan])))
        layers.append(L)
    main_dt = float(rng.choice([600., 0.0, -1.0, np.nan, 1e-9, 1e12]))
    max_dx = float(rng.choice([0.04, 0.0, -1.0, np.nan, 1e-12]))
    min_dt = float(rng.choice([60., 0.0, -1.0, np.nan, 1e9]))
    signal.alarm(20)
    try:
        binding.discretize(layers, main_dt, max_dx, min_dt, 1., float(rng.choice([0.7, np.nan, 1e9])))
    except TimeoutError:
        bad += 1
        print("FAIL damaged construction hangs the setup library: %r main_dt %r max_dx %r min_dt %r" % (layers, main_dt, max_dx, min_dt), flush=True)
    except MemoryError:
        n_refused += 1
    except Exception:  # noqa
        n_refused += 1
    finally:
        signal.alarm(0)
    n_dmg += 1
print("fuzz_setup: %d damaged constructions answered (%d with an error), none hung or crashed" % (n_dmg, n_refused) if not bad else "fuzz_setup: FAILURES above")

# the model builder (ThermalModel::new, model.rs:215-354) with damaged zones, surfaces and constructions: a model — which the
# planner then takes or refuses — or an error code
from heat_amd import ModelBuilder
n_built = n_ref = 0
for k in range(1500):
    rng = np.random.default_rng(seed * 7919 + k)
    try:
        mb = ModelBuilder(int(rng.choice([20, 1, 60, 0, -3, 10**6])), int(rng.choice([-1, 0, 1, 2, 3, 5, 99, -7])))
    except binding.HeatError:
        n_ref += 1
        continue
    try:
        for _ in range(int(rng.integers(0, 4))):
            mb.add_zone(float(rng.choice([600., 0.0, -1.0, np.nan, 1e300])))
        for _ in range(int(rng.integers(0, 6))):
            layers = []
            for i in range(int(rng.integers(0, 4))):
                if rng.random() < 0.2:
                    layers.append(dict(thickness=float(rng.choice([0.02, 0.0, np.nan])), is_gas=True, gas=int(rng.choice([mdl.AIR, 9]))))
                else:
                    layers.append(dict(thickness=float(rng.choice([0.2, 0.02, 0.0, -1., np.nan, 1e6])), k=float(rng.choice([0.8, 0.0, np.nan])),
                                       rho=float(rng.choice([1700., 0.0, -1.])), cp=float(rng.choice([800., np.nan, 0.0]))))
            mb.add_surface(layers, float(rng.choice([60., 0.0, -1., np.nan])), float(rng.choice([46., 0.0, np.nan])),
                           [float(rng.choice([0., 1., np.nan])), float(rng.choice([-1., 0.])), float(rng.choice([0., 1., 2.]))],
                           float(rng.choice([1.5, -1., np.nan, 1e9])), int(rng.choice([0, 1, 2, 3, -1, 9])), int(rng.choice([0, 1, 2, 3, -1])),
                           front_zone=int(rng.choice([0, 1, 5, -1])), back_zone=int(rng.choice([0, 2, -3])),
                           is_fenestration=bool(rng.random() < 0.3))
        md_b, _, _ = mb.finish()
        try:
            binding.plan_check(md_b)
        except binding.HeatError:
            pass
        n_built += 1
    except binding.HeatError:
        n_ref += 1
    finally:
        mb.close()
print("fuzz_setup: model builder: %d damaged models refused, %d built, no crash" % (n_ref, n_built))
sys.exit(1 if bad else 0)
