"""Differential fuzzing of the HIP path against the CPU oracle (test infrastructure, like tests/): random models from the
synthetic generators with random rewiring, sizes on both sides of the planner's thresholds (8192 surfaces, a workgroup, a
team), random splits of the march into calls, random planner modes / blocking factors / graph replay — every owned slot at
1e-9 against the oracle, pass counts of the no-mass loop equal, unowned slots untouched.
    python tools/fuzz.py [SECONDS] [FIRST_SEED]          python tools/fuzz.py SECONDS 0 SEED [SEED ...]   (those cases again)
Prints one line per case; a failing case prints its recipe (seed, options) and the worst slot, and the run goes on.
Round 3: 4 400 cases in seven minutes found two faults of the teams of workgroups — the exchange areas still being zeroed on
the null stream while the first team launch published into them, and members of more than sixteen zones waiting for each
other's sums zone by zone — both fixed, their seeds kept as tests (tests/test_parity_gpu.py).""" 
import faulthandler, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from heat_amd import HeatBatch, modeldict as mdl
from oracle import oracle
from test_parity_gpu import assert_state_close



def oracle_march(md, state, w, a0, b0):
    """The oracle's march; a large model on all host cores (the threaded march does not count passes: counted as equal
    to whatever the GPU says — the states still have to agree)."""
    om = oracle.OracleModel(md)
    if int(md["n_surfaces"]) < 60_000:
        return om.march(state, w, a0, b0)
    rc, _ = om.march(state, w, a0, b0, threads=os.cpu_count() or 8)
    return rc, None


class Discontinuity(Exception):
    """The case sits on a discontinuity of the reference's own algorithm: see check()."""


class PassCount(Exception):
    """The states agree at 1e-9, the pass counts of the no-mass loops do not."""


class IllConditioned(Exception):
    """Temperatures agree at 1e-9; a convection coefficient of a side a few millikelvin off its air differs beyond that."""


def check(md, ref, got, gpu_iters, iters, oracle_again):
    """Every owned slot at 1e-9, pass counts equal — unless the ORACLE itself moves by more than that when its long-wave inputs
    move by one part in 1e15: the no-mass loop leaves by `err > old_err` (surface.rs:842-848) or by its tolerance, and an
    iteration that stagnates leaves a pass earlier or later on the last bit of a sum — a jump of the size of the loop's
    tolerance (1e-5 K) in every wall of the zone. No implementation can agree with another to 1e-9 there (the Rust reference
    and this C oracle would not); such a case is reported as what it is and not counted as a difference."""
    try:
        assert_state_close(md, ref, got)
        if iters is not None and gpu_iters != iters:
            # every slot agrees at 1e-9 and the passes of the no-mass loops do not: a loop that has converged to the last bits
            # leaves by `err > old_err` a pass earlier or later (the libraries' pow differs in the last bit) — the pass it
            # adds or drops changes nothing that shows at 1e-9
            raise PassCount("no-mass passes %d on the GPU, %d in the oracle; every slot within 1e-9" % (gpu_iters, iters))
    except AssertionError as e:
        slots = mdl.node_slots(md)
        # A convection coefficient is C |dT|^(1/3) (convection.rs:87-110): where air and surface are a few millikelvin apart
        # it moves by a relative 1e-7 for a face temperature that moves by 1e-10 K — inside the tolerance the temperatures
        # are held to (seed 121815: a window 5.9 mK off its zone, face temperatures 1.4e-10 K apart, hs 4e-8 relative).
        # Node and zone temperatures at 1e-9, coefficients and flows at 1e-6: ill-conditioned, not different.
        if str(e).startswith(("hs_", "flow_")):
            tight = all(np.all(np.abs(ref[i] - got[i]) <= 1e-9 * (1.0 + np.abs(ref[i]))) for i in (slots, md["zone_slot"]))
            loose = all(np.all(np.abs(ref[md[k]] - got[md[k]]) <= 1e-6 * (1e-3 + np.abs(ref[md[k]])))
                        for k in ("hs_front_slot", "hs_back_slot", "flow_front_slot", "flow_back_slot"))
            if tight and loose:
                raise IllConditioned(str(e)[:160])
        for eps in (1e-15, -1e-15, 3e-15, -4e-15):
            r2 = oracle_again(eps)
            jump = float(np.max(np.abs(r2[slots] - ref[slots])))
            if jump > 1e-9 * (1.0 + float(np.max(np.abs(ref[slots])))):
                raise Discontinuity("oracle moves by %.3e K for a relative %.0e on its long-wave inputs (GPU vs oracle: %s)" % (jump, eps, str(e)[:120]))
        raise


def make_case(seed):
    """The case of a seed: (model, state, generator name, weather, a0, b0, batch options, call boundaries)."""
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 7))
    big = rng.random() < 0.25
    S = int(rng.integers(9000, 40000)) if big else int(rng.integers(30, 1500))
    # one case in thirty is large: more tiles than the persistent grids have wavefronts (several tiles per wavefront,
    # balanced rounds, the descriptor prefetch, zig-zag sweeps over real lists), teams in several rounds
    huge = np.random.default_rng(seed ^ 0xb16).random() < 1.0 / 30.0
    if huge:
        S = int(np.random.default_rng(seed ^ 0xb17).integers(100_000, 400_000))
    if kind == 0:
        Z = int(rng.integers(2, max(3, S // 20)))
        md, st = mdl.clustered_massive(S, Z=Z, dt=45.0, seed=seed)
        name = "clustered_massive"
    elif kind == 1:
        Z = int(rng.integers(2, max(3, S // 20)))
        md, st = mdl.rooms_with_windows(S, Z=Z, dt=45.0, seed=seed)
        name = "rooms_with_windows"
    elif kind == 2:
        Z = int(rng.integers(1, max(2, S // 50)))
        md, st = mdl.ragged_mixed(S, Z=Z, dt=45.0, seed=seed, n_lo=int(rng.integers(2, 9)), n_hi=int(rng.integers(9, 65)))
        name = "ragged_mixed"
    elif kind == 3:
        Z = int(rng.integers(1, max(2, S // 60)))
        md, st = mdl.glazing_cavity(S, Z=Z, dt=45.0, seed=seed, trombe_fraction=float(rng.uniform(0, 1)))
        name = "glazing_cavity"
    elif kind == 4:
        rooms = int(rng.choice([3, 8, 20, 24, 40, 64, 100]))
        n = int(rng.choice([7, 13, 16, 20, 32]))
        md, st = mdl.partitioned_buildings(S, n, rooms=rooms, dt=45.0, seed=seed)
        name = "partitioned_buildings(rooms=%d, n=%d)" % (rooms, n)
    elif kind == 5:
        n = int(rng.integers(2, 65))
        md, st = mdl.uniform_massive(S, n, Z=max(1, S // int(rng.integers(20, 200))), dt=45.0, seed=seed)
        name = "uniform_massive(n=%d)" % n
    else:
        md, st = mdl.uniform_massive(S, 20, Z=max(1, S // 100), dt=90.0, identical=True, vertical=True)
        name = "identical"
    S = int(md["n_surfaces"]); Z = int(md["n_zones"])
    # rewiring, as the planner stress test does
    if rng.random() < 0.6 and Z > 1:
        pick = rng.random(S)
        both = (md["front_kind"] == mdl.SPACE) & (md["back_kind"] == mdl.SPACE)
        rew = both & (pick < rng.uniform(0, 0.3))
        md["front_zone"] = np.where(rew, rng.integers(0, Z, S), md["front_zone"]).astype(np.int32)
        same = both & (pick > 0.93)
        md["front_zone"] = np.where(same, md["back_zone"], md["front_zone"]).astype(np.int32)
        nodes = np.diff(md["node_offset"])
        lone = (pick > 0.5) & (pick < 0.54) & (nodes > 4)
        md["front_kind"] = np.where(lone, mdl.AMBIENT, md["front_kind"]).astype(np.int32)
        md["back_kind"] = np.where(lone, mdl.OUTDOOR, md["back_kind"]).astype(np.int32)
        md["front_ambient"] = np.where(lone, 12.5, md["front_ambient"])
    mdl.perturb_initial_temperatures(md, st, rng)
    n_sub = int(rng.integers(1, 14))
    if huge:
        n_sub = min(n_sub, 4)
    elif not big and np.random.default_rng(seed ^ 0x10a6).random() < 0.06:
        # a long march (small models only): more sub-timesteps than one captured graph holds (32), blocks and a remainder
        n_sub = int(np.random.default_rng(seed ^ 0x10a7).integers(33, 90))
    w = mdl.weather_series(n_sub, float(md["dt"]), wind_speed=float(rng.uniform(0.0, 8.0)), wind_deg=float(rng.uniform(0, 360)))
    a0 = rng.uniform(0., 50., Z)
    b0 = rng.uniform(0., 2., Z)
    modes = [dict(), dict(fuse_always=True), dict(no_fusion=True)]
    kw = dict(modes[int(rng.integers(0, 3))])
    if rng.random() < 0.3:
        kw["nodes_per_lane"] = int(rng.choice([4, 8, 16]))
    if rng.random() < 0.15:
        kw["no_palette"] = True
    kw["use_graph"] = bool(rng.random() < 0.5)
    # (the draws below came later: earlier seeds keep their cases)
    rng2 = np.random.default_rng(seed ^ 0xf1c5)
    if rng2.random() < 0.10:
        # the reference's debug-only overrides of the convection coefficients (surface.rs:374-380), on some sides
        for key in ("front_hs_fix", "back_hs_fix"):
            v = np.full(S, np.nan)
            on = rng2.random(S) < 0.2
            v[on] = rng2.uniform(1., 10., int(on.sum()))
            md[key] = v
    if rng2.random() < 0.05:
        kw["force_general"] = True
    cuts = sorted(set(int(c) for c in rng.integers(1, n_sub + 1, int(rng.integers(0, 4)))) | {n_sub})
    return md, st, name, w, a0, b0, kw, cuts


def run_case(seed):
    """Marches the case on the GPU and on the oracle; returns a line of text, raises on any difference."""
    md, st, name, w, a0, b0, kw, cuts = make_case(seed)
    ref = st.copy()
    rc, iters = oracle_march(md, ref, w, a0, b0)
    if rc != 0:
        return None

    def oracle_again(eps):  # the same march with the long-wave inputs moved by a relative eps
        r2 = st.copy()
        for key in ("ir_front_slot", "ir_back_slot"):
            r2[md[key]] *= 1.0 + eps
        r2[mdl.node_slots(md)] *= 1.0 + eps
        oracle.OracleModel(md).march(r2, w, a0, b0)
        return r2
    got = st.copy()
    rng3 = np.random.default_rng(seed ^ 0x5e7f)
    ts = None
    if rng3.random() < 0.12:  # (a caller's stream — torch's, as heat_amd/sharded.py passes it — instead of the batch's own)
        try:
            import torch
            ts = torch.cuda.Stream()
            kw = dict(kw, stream=ts.cuda_stream)
        except Exception:  # noqa
            ts = None
    with HeatBatch(md, **kw) as b:
        b.upload_state(got)
        lo = 0
        for c in cuts:
            if c > lo:
                if rng3.random() < 0.2:
                    b.set_fusion(bool(rng3.random() < 0.5))   # (heat_batch_set_fusion between two calls)
                b.march_resident(w[lo:c], a0, b0)
                lo = c
        b.synchronize()
        b.download_state(got)
        gpu_iters = b.nomass_iterations()
        info = "classes %s fused %d launches %d" % (b.class_counts(), b.n_fused_surfaces, b.n_fused_launches)
    check(md, ref, got, gpu_iters, iters, oracle_again)
    return "%-40s S=%-6d Z=%-5d n_sub=%-2d calls %s %s | %s" % (name, int(md["n_surfaces"]), int(md["n_zones"]), len(w), cuts, kw, info)


def run_dropin_case(seed):
    """The drop-in call as a host uses it: a caller-owned state, between the march calls "other modules" write new solar and
    long-wave irradiances (and every call brings its own zone terms a0 / b0), heat_batch_march uploads the inputs, marches and
    downloads the outputs the call asks for (a random mask; the rest is fetched at the end). The oracle marches call by call
    on the same inputs."""
    md, st, name, w, a0, b0, kw, cuts = make_case(seed)
    rng = np.random.default_rng(seed ^ 0xd20b)
    Z = int(md["n_zones"])
    om = oracle.OracleModel(md)
    ref = st.copy()
    got = st.copy()
    total = 0
    masks = []
    script = []  # (inputs written, weather slice, a0, b0) per call, for oracle_again
    with HeatBatch(md, **kw) as b:
        b.upload_state(got)
        lo = 0
        for c in cuts:
            if c <= lo:
                continue
            # what the solar and long-wave modules would write between two timesteps
            written = {}
            for key, hi in (("solar_front_slot", 700.), ("solar_back_slot", 200.), ("ir_front_slot", 450.), ("ir_back_slot", 450.)):
                if rng.random() < 0.7:
                    v = rng.uniform(0., hi, int(md["n_surfaces"]))
                    ref[md[key]] = v
                    got[md[key]] = v
                    written[key] = v
            a_c = a0 * rng.uniform(0.5, 1.5)
            b_c = b0 * rng.uniform(0.5, 1.5)
            script.append((written, w[lo:c], a_c, b_c))
            rc, it = om.march(ref, w[lo:c], a_c, b_c)
            if rc != 0:
                return None
            total += it
            mask = int(rng.choice([HeatBatch.OUT_ALL, HeatBatch.OUT_ALL, HeatBatch.OUT_SCALARS | HeatBatch.OUT_ZONES, HeatBatch.OUT_ZONES, HeatBatch.OUT_NODES]))
            masks.append(mask)
            b.march(got, w[lo:c], a_c, b_c, outputs=mask)
            lo = c
        b.synchronize()
        b.download_outputs(got, HeatBatch.OUT_ALL)
        gpu_iters = b.nomass_iterations()
        info = "classes %s fused %d launches %d" % (b.class_counts(), b.n_fused_surfaces, b.n_fused_launches)

    def oracle_again(eps):
        r2 = st.copy()
        r2[mdl.node_slots(md)] *= 1.0 + eps
        o2 = oracle.OracleModel(md)
        for written, wv, a_c, b_c in script:
            for key, v in written.items():
                r2[md[key]] = v * (1.0 + eps) if key.startswith("ir") else v
            o2.march(r2, wv, a_c, b_c)
        return r2
    check(md, ref, got, gpu_iters, total, oracle_again)
    return "%-40s S=%-6d Z=%-5d n_sub=%-2d drop-in calls %s outputs %s %s | %s" % (
        name, int(md["n_surfaces"]), Z, len(w), cuts, masks, kw, info)


def run_sharded_case(seed):
    """The same case cut into 2-8 shards that all live on this one device: heat_partition's cut (whole clusters per rank, an
    oversized cluster cut through) or — every other time — arbitrary surface ranges (many zones shared), each shard a batch of
    its own, the zones they share exchanged through the split-phase ABI (heat_batch_step_surfaces -> every rank's partial
    (a, b) block -> heat_batch_step_zones, the blocks summed in rank order) with this process playing the collective."""
    import torch
    from heat_amd import binding
    from heat_amd.sharded import zone_roles
    md, st, name, w, a0, b0, kw, cuts = make_case(seed)
    rng = np.random.default_rng(seed ^ 0x5eed)
    S = int(md["n_surfaces"]); Z = int(md["n_zones"])
    R = int(rng.choice([2, 3, 4, 8]))
    if S < 2 * R:
        return None
    if rng.random() < 0.5:
        ranks, _ = binding.partition(md, R)
        how = "heat_partition"
    else:
        edges = np.sort(rng.choice(np.arange(1, S), R - 1, replace=False))
        ranks = np.searchsorted(edges, np.arange(S), side="right").astype(np.int32)
        how = "ranges"
    if len(np.unique(ranks)) < R:
        return None
    ref = st.copy()
    rc, iters = oracle.OracleModel(md).march(ref, w, a0, b0)
    if rc != 0:
        return None
    kw = {k: v for k, v in kw.items() if k != "use_graph"}
    batches = [HeatBatch(md, n_ranks=R, rank=r, rank_of_surface=ranks, **kw) for r in range(R)]
    try:
        masks = [b.touched_zones() for b in batches]
        cnt = np.sum(np.asarray(masks, dtype=np.int64), axis=0)
        shared = None
        for r, b in enumerate(batches):
            shared, owned = zone_roles(cnt, masks[r], r, R)
            b.set_owned_zones(owned)
            b.set_shared_zones(shared)
        n = max(2 * len(shared), 2)
        gathered = torch.zeros(R * n, dtype=torch.float64, device="cuda")
        for r, b in enumerate(batches):
            b.use_partials(gathered.data_ptr() + r * n * 8)
            b.upload_state(st)
        lo = 0
        for c in cuts:
            if c <= lo:
                continue
            wv = w[lo:c]
            for b in batches:
                b.set_weather(wv, a0, b0)
            for i in range(len(wv)):
                for b in batches:
                    b.step_surfaces(i)
                torch.cuda.synchronize()
                if len(shared):
                    for b in batches:
                        b.step_zones(gathered.data_ptr(), R)
                    torch.cuda.synchronize()
            lo = c
        got = st.copy()
        total = 0
        for b in batches:
            b.synchronize()
            b.download_state(got)
            total += b.nomass_iterations()
        info = "fused %s" % [b.n_fused_surfaces for b in batches]
    finally:
        for b in batches:
            b.close()

    def oracle_again(eps):
        r2 = st.copy()
        for key in ("ir_front_slot", "ir_back_slot"):
            r2[md[key]] *= 1.0 + eps
        r2[mdl.node_slots(md)] *= 1.0 + eps
        oracle.OracleModel(md).march(r2, w, a0, b0)
        return r2
    check(md, ref, got, total, iters, oracle_again)
    return "%-40s S=%-6d Z=%-5d n_sub=%-2d %d shards by %s, %d zones shared %s | %s" % (
        name, S, Z, len(w), R, how, len(shared), kw, info)


if __name__ == "__main__":
    try:  # (torch carries the sharded cases' exchange buffer: it wants to see the device before anybody else holds it)
        import torch
        torch.cuda.is_available() and torch.zeros(1, device="cuda")
    except Exception:  # noqa
        pass
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    only = [int(a) for a in sys.argv[3:]]          # given: exactly these seeds (a failing case again)
    oracle.lib()
    t_end = time.time() + budget
    n_ok = n_bad = n_disc = n_cond = n_pass = 0
    seeds = iter(only) if only else iter(range(seed0, 1 << 62))
    last = seed0
    for seed in seeds:
        if time.time() >= t_end:
            break
        last = seed
        print("     seed %d ..." % seed, flush=True)          # (a case that hangs or crashes the process names itself)
        faulthandler.dump_traceback_later(90, exit=True)       # ... and ends the run instead of the GPU box's patience
        try:
            line = run_sharded_case(seed) if seed % 4 == 3 else (run_dropin_case(seed) if seed % 4 == 1 else run_case(seed))
            if line is not None:
                n_ok += 1
                print("ok   seed %d %s" % (seed, line), flush=True)
        except Discontinuity as e:
            n_disc += 1
            print("disc seed %d: %s" % (seed, e), flush=True)
        except IllConditioned as e:
            n_cond += 1
            print("cond seed %d: %s" % (seed, e), flush=True)
        except PassCount as e:
            n_pass += 1
            print("pass seed %d: %s" % (seed, e), flush=True)
        except Exception as e:  # noqa
            n_bad += 1
            try:
                kw = make_case(seed)[6]
            except Exception:  # noqa
                kw = None
            print("FAIL seed %d %s: %s" % (seed, kw, "".join(traceback.format_exception_only(type(e), e)).strip()[:600]), flush=True)
    print("fuzz: %d ok, %d failed, %d on a discontinuity of the reference's algorithm, %d with an ill-conditioned coefficient, "
          "%d with equal states and other pass counts, seeds %d..%d" % (n_ok, n_bad, n_disc, n_cond, n_pass, only[0] if only else seed0, last))
    sys.exit(1 if n_bad else 0)
