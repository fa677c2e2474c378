"""Parity of the HIP path (through the C ABI, heat_amd/lib/libheat_amd.so) against the CPU oracle.

Tolerance: BASELINE.json's north_star asks for node temperatures within 1e-9 relative of the
reference CPU path. Every comparison below uses rtol = 1e-9 with atol = 1e-9 (temperatures are in
Celsius and cross zero, heat flows reach zero at equilibrium).
"""
import math
import os

import numpy as np
import pytest

from helpers import BRICKWORK, POLYURETHANE, surfaces_model
from heat_amd import HeatBatch, HeatError
from heat_amd import modeldict as mdl

pytestmark = pytest.mark.gpu

RTOL = 1e-9
ATOL = 1e-9


def run_both(oracle, md, state0, weather, a0=None, b0=None, **opts):
    ref = state0.copy()
    rc, iters = oracle.OracleModel(md).march(ref, weather, a0, b0)
    assert rc == 0
    got = state0.copy()
    with HeatBatch(md, **opts) as b:
        b.upload_state(got)
        b.march(got, weather, a0, b0)
        counts = b.class_counts()
        gpu_iters = b.nomass_iterations()
        n_fused = b.n_fused_surfaces
        # (a planned cluster-resident march must really have run: calls of >= 2 sub-timesteps, or any in a small batch)
        assert (b.n_fused_launches > 0) == (n_fused > 0 and (len(weather) >= 2 or md["n_surfaces"] <= 8192))
    if n_fused > 0 and "no_fusion" not in opts:
        # the planner sent (part of) this batch through the cluster-resident march: the streamed kernels of the
        # same batch are held to the same oracle
        streamed = state0.copy()
        with HeatBatch(md, **dict(opts, no_fusion=True, fuse_always=False)) as b:
            b.upload_state(streamed)
            b.march(streamed, weather, a0, b0)
            assert b.nomass_iterations() == iters
        assert_state_close(md, ref, streamed)
    return ref, got, iters, gpu_iters, counts


def assert_state_close(md, ref, got):
    for name, idx in (("nodes", mdl.node_slots(md)), ("hs_front", md["hs_front_slot"]), ("hs_back", md["hs_back_slot"]),
                      ("flow_front", md["flow_front_slot"]), ("flow_back", md["flow_back_slot"]),
                      ("zones", md["zone_slot"])):
        r, g = ref[idx], got[idx]
        err = np.abs(r - g) / (ATOL + RTOL * np.abs(r))
        assert np.all(np.isfinite(g)), name
        assert err.max() <= 1.0 if len(err) else True, "%s: worst |diff| %.3e at %d (ref %.17g, got %.17g)" % (
            name, np.abs(r - g).max(), int(err.argmax()), r[err.argmax()], g[err.argmax()])
    # slots the path does not own must be untouched
    owned = np.zeros(len(ref), dtype=bool)
    for idx in (mdl.node_slots(md), md["hs_front_slot"], md["hs_back_slot"], md["flow_front_slot"],
                md["flow_back_slot"], md["zone_slot"]):
        owned[idx] = True
    assert np.array_equal(ref[~owned], got[~owned], equal_nan=True)


@pytest.mark.parametrize("npl,no_palette", [(0, False), (4, False), (8, False), (16, False), (0, True), (4, True),
                                            (8, True), (16, True)])
def test_config2_identical_massive_walls(oracle, npl, no_palette):
    # BASELINE config 2 at reduced S: identical 3-layer massive walls x 20 nodes, RK4 only
    md, st = mdl.uniform_massive(300, 20, Z=3, dt=90.0, identical=True, vertical=True)
    w = mdl.weather_series(60, 90.0)
    ref, got, _, _, counts = run_both(oracle, md, st, w, nodes_per_lane=npl, no_palette=no_palette)
    assert counts[3] + counts[4] == 0  # all on the fast path
    assert_state_close(md, ref, got)


@pytest.mark.parametrize("n,npl", [(32, 0), (32, 4), (32, 8), (7, 0), (13, 0), (64, 4), (33, 8), (2, 0), (50, 16)])
def test_uniform_massive_random_materials(oracle, n, npl):
    md, st = mdl.uniform_massive(517, n, Z=5, dt=45.0, seed=n * 7 + npl)
    w = mdl.weather_series(40, 45.0, wind_speed=4.5, wind_deg=200.0)
    ref, got, _, _, counts = run_both(oracle, md, st, w, nodes_per_lane=npl)
    assert counts[3] == 0
    assert_state_close(md, ref, got)


def test_general_kernel_matches_fast_kernel_and_oracle(oracle):
    md, st = mdl.uniform_massive(200, 24, Z=4, dt=45.0, seed=3)
    w = mdl.weather_series(30, 45.0)
    ref, got, _, _, counts = run_both(oracle, md, st, w, force_general=True)
    assert counts[4] == 200
    assert_state_close(md, ref, got)


def test_config3_ragged_mixed(oracle):
    # BASELINE config 3 at reduced S: ragged 8..64 nodes, massive / mixed / pure no-mass, all boundary kinds
    md, st = mdl.ragged_mixed(3000, Z=30, dt=45.0, seed=20260401)
    w = mdl.weather_series(25, 45.0)
    a0 = np.linspace(0., 500., 30)
    b0 = np.linspace(0., 20., 30)
    ref, got, iters, gpu_iters, counts = run_both(oracle, md, st, w, a0, b0)
    assert counts[3] > 0 and sum(counts[:3]) > 0  # small no-mass walls + fast classes
    assert iters == gpu_iters, "no-mass loop took a different number of passes (%d vs %d)" % (iters, gpu_iters)
    assert_state_close(md, ref, got)


def test_config5_glazing_and_cavities(oracle):
    md, st = mdl.glazing_cavity(400, Z=8, dt=45.0)
    w = mdl.weather_series(25, 45.0)
    ref, got, iters, gpu_iters, counts = run_both(oracle, md, st, w)
    assert counts[3] > 0 and sum(counts[:3]) > 0 and counts[4] == 0  # glazing: small kernel; Trombe: fast path with cavity
    assert iters == gpu_iters
    assert_state_close(md, ref, got)


def test_config2_and_config5_at_baseline_size(oracle):
    """BASELINE.json's configs 2 and 5 at their full sizes against the oracle (it runs them in seconds on the host
    cores): 10 000 identical 3-layer walls x 20 nodes x 200 sub-timesteps; 100 000 double glazings + 100 000
    Trombe-like walls x 25 sub-timesteps. The planner's choice and the streamed kernels both."""
    md, st = mdl.uniform_massive(10_000, 20, Z=100, dt=90.0, identical=True, vertical=True)
    w = mdl.weather_series(200, 90.0)
    ref = st.copy()
    rc, iters = oracle.OracleModel(md).march(ref, w, threads=16)
    assert rc == 0
    for kw in (dict(), dict(no_fusion=True)):
        got = st.copy()
        with HeatBatch(md, use_graph=True, **kw) as b:
            b.upload_state(got)
            for i in range(0, 200, 20):                 # ten march calls of 20 sub-timesteps
                b.march_resident(w[i:i + 20])
            b.synchronize()
            b.download_state(got)
        assert_state_close(md, ref, got)
    md, st = mdl.glazing_cavity(200_000, Z=2000, dt=45.0)
    w = mdl.weather_series(25, 45.0)
    ref = st.copy()
    rc, iters = oracle.OracleModel(md).march(ref, w, threads=16)
    assert rc == 0
    got = st.copy()
    with HeatBatch(md) as b:
        b.upload_state(got)
        b.march(got, w)
        counts = b.class_counts()
        # (the threaded oracle does not count the no-mass passes; test_config5_glazing_and_cavities pins them)
    assert counts[3] > 90_000 and sum(counts[:3]) > 90_000 and counts[4] == 0
    assert_state_close(md, ref, got)


@pytest.mark.parametrize("npl", [0, 4, 8, 16])
def test_nomass_chunks_inside_the_wall_and_of_two_nodes_on_the_fast_path(oracle, npl):
    """ThermalSurfaceData::march solves every no-mass chunk wherever it sits (surface.rs:950-965); get_chunks
    (discretization.rs:144-160) makes them of any two adjacent light layers: render on insulation at a face (two
    nodes), an insulation layer and an air gap inside a cavity wall (one node between two massive chunks). These run
    in the register kernel — nothing is left to the one-lane-per-surface catch-all — with the oracle's pass counts."""
    conc = dict(thickness=0.2, k=0.816, rho=1700., cp=800.)
    brick = dict(thickness=0.1, k=0.6, rho=1600., cp=840.)
    poly = dict(thickness=0.02, k=0.0252, rho=17.5, cp=2400.)
    wool = dict(thickness=0.05, k=0.04, rho=30., cp=1000.)
    walls = {"render on insulation outside": [poly, wool, conc],
             "insulation and lining inside": [conc, wool, poly],
             "cavity wall": [brick, wool, poly, brick],
             "both": [poly, wool, conc, wool, poly],
             "facing and cavity": [poly, brick, poly, wool, conc, poly]}   # (more than three conductances: the wide palette)
    mds, states = [], []
    for name, lay in walls.items():
        lay = [dict(L, front_thermal_abs=0.2, back_thermal_abs=0.2, front_solar_abs=0.6, back_solar_abs=0.6) for L in lay]
        d = oracle.discretize(lay, 180., 0.04, 60., 1., math.pi / 2)
        _, nomass = oracle.get_chunks(d["mass"])
        assert nomass and max(e - i for i, e in nomass) <= 2, (name, nomass)
        assert any(e - i == 2 or (i > 0 and e < d["n_nodes"]) for i, e in nomass), (name, nomass)   # not just one-node facings
        md, st = surfaces_model(d, 180. / d["tstep_subdivision"] / 2., mdl.OUTDOOR, mdl.SPACE, n_zones=2, zone_volume=[300., 200.],
                                back_zone=1, front_emis=0.2, back_emis=0.2, area=12., perimeter=14., cos_tilt=0.0,
                                normal=(0.6, -0.8, 0.), copies=37)
        mds.append(md); states.append(st)
    dt = min(m["dt"] for m in mds)
    # one model of all of them (same dt), every second wall between the two zones
    md = mdl.empty(sum(m["n_surfaces"] for m in mds), 2, dt)
    off = [0]
    for m in mds:
        off.extend((off[-1] + np.asarray(m["node_offset"][1:])).tolist())
    md["node_offset"] = np.asarray(off, dtype=np.int64)
    for k in ("mass", "uvalue", "front_alpha", "back_alpha"):
        md[k] = np.concatenate([m[k] for m in mds])
    for k in mdl.PER_SURFACE_F64 + mdl.PER_SURFACE_I32:
        md[k] = np.concatenate([m[k] for m in mds])
    S = md["n_surfaces"]
    rng = np.random.default_rng(5 + npl)
    sp = np.arange(S) % 2 == 1
    md["front_kind"] = np.where(sp, mdl.SPACE, mdl.OUTDOOR).astype(np.int32)
    md["front_zone"] = np.zeros(S, dtype=np.int32)
    md["cos_tilt"] = rng.choice([0.0, 0.707, -1.0], S)
    md["zone_volume"] = np.array([300., 200.])
    st = mdl.layout_state(md)
    mdl.perturb_initial_temperatures(md, st, rng)
    st[md["solar_front_slot"]] = rng.uniform(0., 600., S)
    st[md["solar_back_slot"]] = rng.uniform(0., 50., S)
    mdl.set_ir_from_air(md, st, 8.0)
    w = mdl.weather_series(30, dt, wind_speed=4.0, wind_deg=40.0)
    ref, got, iters, gpu_iters, counts = run_both(oracle, md, st, w, np.array([40., 0.]), np.array([1., 0.5]),
                                                  nodes_per_lane=npl)
    assert counts[4] == 0 or npl != 0, counts          # the planner's own choice leaves nothing to the catch-all
    if npl != 0:
        assert counts[4] < S                            # (a forced blocking factor may cut a two-node chunk in two)
    assert iters == gpu_iters and iters > 0
    assert_state_close(md, ref, got)
    if npl in (0, 8):
        # ... and in the cluster-resident march (the two zones are one cluster of every wall; 8 or 4 nodes per lane carry
        # the chunk loop there): pass counts and results as the oracle's
        # (half the walls: all of them are more wavefronts than a workgroup holds)
        half = mdl.subset(md, np.sort(np.concatenate([np.arange(0, S, 4), np.arange(1, S, 4)])))
        a0h, b0h = np.array([40., 0.]), np.array([1., 0.5])
        ref_h = st.copy()
        rc, iters_h = oracle.OracleModel(half).march(ref_h, w, a0h, b0h)
        assert rc == 0 and iters_h > 0
        fused = st.copy()
        with HeatBatch(half, nodes_per_lane=npl, fuse_always=True) as b:
            assert b.n_fused_surfaces == half["n_surfaces"], (b.n_fused_surfaces, b.class_counts())
            b.upload_state(fused)
            b.march(fused, w[:11], a0h, b0h)
            b.march(fused, w[11:], a0h, b0h)
            assert b.n_fused_launches >= 2 and b.nomass_iterations() == iters_h
        for idx in (mdl.node_slots(half), half["hs_front_slot"], half["hs_back_slot"], half["flow_front_slot"],
                    half["flow_back_slot"], half["zone_slot"]):
            assert np.allclose(fused[idx], ref_h[idx], rtol=RTOL, atol=ATOL)


def test_reference_unit_test_walls_through_the_abi(oracle):
    # test_march_massive_1 / test_march_nomass (surface.rs:1087-1443) with the debug hs overrides
    lay = [dict(thickness=20. / 1000., **BRICKWORK)]
    d = oracle.discretize(lay, 300.0, 0.01, 1.0, 1., 0.)
    md, st = surfaces_model(d, 300.0 / d["tstep_subdivision"], mdl.OUTDOOR, mdl.OUTDOOR, hs_fix=(10., 10.), copies=3)
    st[md["ir_front_slot"]] = mdl.SIGMA * 283.15 ** 4
    st[md["ir_back_slot"]] = mdl.SIGMA * 283.15 ** 4
    w = np.tile([10., 0., 0.], (400, 1))
    ref, got, *_ = run_both(oracle, md, st, w)
    assert_state_close(md, ref, got)
    assert np.all(np.abs(got[mdl.node_slots(md)] - 10.0) < 0.5)

    th = 3. / 1000.
    d = oracle.discretize([dict(thickness=th, **POLYURETHANE)] * 2, 3.0, th / 7., 10.0, 1., 0.)
    md, st = surfaces_model(d, 3.0 / d["tstep_subdivision"], mdl.OUTDOOR, mdl.OUTDOOR, hs_fix=(10., 10.), copies=2)
    ref, got, iters, gpu_iters, _ = run_both(oracle, md, st, np.tile([10., 0., 0.], (3, 1)))
    assert iters == gpu_iters
    assert_state_close(md, ref, got)


def test_march_equals_resident_march_plus_download(oracle):
    md, st = mdl.ragged_mixed(500, Z=5, dt=45.0, seed=11)
    w = mdl.weather_series(10, 45.0)
    a = st.copy()
    b_ = st.copy()
    with HeatBatch(md) as b:
        b.upload_state(a)
        b.march(a, w)
    with HeatBatch(md, use_graph=True) as b:
        b.upload_state(b_)
        b.march_resident(w[:4])
        b.march_resident(w[4:])
        b.synchronize()
        b.download_state(b_)
    assert np.array_equal(a, b_)  # same kernels, same order: bitwise equal


def test_zero_length_march_is_a_no_op_also_as_the_first_call_of_a_graph_batch(oracle):
    """ThermalModel::march with dt_subdivisions == 0 runs its loop body never (model.rs:369): the call returns at
    once, leaves the state alone, and a graph batch must neither capture an empty graph nor divide by its length."""
    md, st = mdl.ragged_mixed(300, Z=3, dt=45.0, seed=21)
    w = mdl.weather_series(6, 45.0)
    ref = st.copy()
    rc, _ = oracle.OracleModel(md).march(ref, w)
    assert rc == 0
    for kw in (dict(use_graph=True), dict(use_graph=True, no_fusion=True), dict()):
        got = st.copy()
        with HeatBatch(md, **kw) as b:
            b.upload_state(got)
            b.march_resident(w[:0])          # first call: nothing captured yet
            b.synchronize()
            before = got.copy()
            b.download_state(before)
            assert np.array_equal(before[mdl.node_slots(md)], st[mdl.node_slots(md)])
            b.march_resident(w[:4])
            b.march_resident(w[:0])          # between two real calls: the captured graph's length stays valid
            b.march_resident(w[4:])
            b.synchronize()
            b.download_state(got)
            b.march(got.copy(), w[:0])       # the drop-in call of no sub-timestep
        assert_state_close(md, ref, got)


def test_split_phase_steps_equal_fused_march(oracle):
    md, st = mdl.ragged_mixed(400, Z=4, dt=45.0, seed=5)
    w = mdl.weather_series(6, 45.0)
    a = st.copy()
    with HeatBatch(md) as b:
        b.upload_state(a)
        b.march(a, w)
    c = st.copy()
    with HeatBatch(md) as b:
        b.upload_state(c)
        b.set_weather(w)
        for i in range(len(w)):
            b.step_surfaces(i)
            b.step_zones(None, 1)
        b.synchronize()
        b.download_state(c)
    ref = st.copy()
    oracle.OracleModel(md).march(ref, w)
    assert_state_close(md, ref, c)
    assert_state_close(md, a, c)


def test_error_codes(oracle):
    d = oracle.discretize([dict(thickness=0.02, **BRICKWORK)], 300., 0.01, 1.0)
    md, st = surfaces_model(d, 10., mdl.GROUND, mdl.OUTDOOR)
    with pytest.raises(HeatError) as e:
        HeatBatch(md)
    assert e.value.code == -2  # HEAT_E_GROUND_BOUNDARY
    md, st = surfaces_model(d, 10., mdl.OUTDOOR, mdl.OUTDOOR)
    md["uvalue"] = md["uvalue"].copy()
    md["uvalue"][0] = np.nan
    with pytest.raises(HeatError) as e:
        HeatBatch(md)
    assert e.value.code == -3  # HEAT_E_UVALUE_NONE
    md, st = surfaces_model(d, 10., mdl.OUTDOOR, mdl.OUTDOOR)
    st[md["first_node_slot"][0]] = np.nan
    with HeatBatch(md) as b:
        b.upload_state(st)
        with pytest.raises(HeatError) as e:
            b.march(st, np.array([[10., 0., 1.]]))
        assert e.value.code > 0  # numerical failure, like the reference's panic


def test_all_fourteen_energyplus_series_through_the_abi_full_length(oracle):
    """every wall series the reference registers (validate_wall_heat_transfer.rs:792-994 — twelve
    construction x radiation cases, the tilted and the horizontal wall) over its full 7 000 rows (5 000 warm-up +
    2 000 compared) through heat_batch_march on a caller-owned state: the same zone temperatures as the oracle's
    harness at 1e-9 and the same RMSE against EnergyPlus. tools/validation_table.py writes the table."""
    import os
    from test_energyplus_series import CASES, GEOMETRY, GOLD, march_series, single_zone_model
    for case in sorted(CASES):
        layers, emis, sol = CASES[case]
        fx = dict(np.load(os.path.join(GOLD, "wall_%s.npz" % case)))
        md, st, d, n_sub = single_zone_model(oracle, layers, emis, sol, **GEOMETRY.get(case, {}))
        ref = march_series(oracle, md, st.copy(), n_sub, fx, emis)
        got_state = st.copy()
        with HeatBatch(md) as b:
            b.upload_state(got_state)
            got = march_series(oracle, md, got_state, n_sub, fx, emis, march=lambda s, w: b.march(s, w))
        assert np.allclose(got, ref, rtol=RTOL, atol=ATOL), case
        sel = slice(5001, None)
        rm_ref = float(np.sqrt(np.mean((ref[sel] - fx["zone_t"][sel]) ** 2)))
        rm_got = float(np.sqrt(np.mean((got[sel] - fx["zone_t"][sel]) ** 2)))
        assert abs(rm_ref - rm_got) < 1e-8 and rm_got < 0.5, (case, rm_ref, rm_got)


@pytest.mark.parametrize("n_ranks,fusion", [(2, "plan"), (3, "plan"), (3, "stream"), (8, "plan")])
def test_shards_of_the_cluster_partition_march_without_any_exchange(oracle, n_ranks, fusion):
    """BASELINE config 4 as `bench.py --gpus N` runs it, here with every rank's batch on this one GPU: heat_partition
    cuts the model along its zone-connected clusters (no zone shared), heat_batch_create_shard builds each rank's batch
    from the WHOLE model's descriptor, every batch marches on its own (no communicator, no collective) and downloads
    into the one caller state: its surfaces, and the zones it owns — faced zones and, of the zones nobody faces, every
    n_ranks-th one, which still follow a0 / b0 (model.rs:410-423). The union equals the single-process oracle."""
    from heat_amd import binding
    md, st = mdl.clustered_massive(2400, Z=96, dt=45.0, seed=21)
    for z in (5, 40, 77):   # zones nobody faces: their walls move to the pair's other zone
        for key in ("front_zone", "back_zone"):
            md[key] = np.where(md[key] == z, z ^ 1, md[key]).astype(np.int32)
    w = mdl.weather_series(11, 45.0, wind_speed=2.5, wind_deg=310.0)
    a0 = np.linspace(5., 80., 96)
    b0 = np.linspace(0.2, 3., 96)
    ref = st.copy()
    rc, iters = oracle.OracleModel(md).march(ref, w, a0, b0)
    assert rc == 0
    ranks, n_shared = binding.partition(md, n_ranks)
    assert n_shared == 0 and ranks.max() == n_ranks - 1
    got = st.copy()
    total_iters = 0
    for r in range(n_ranks):
        with HeatBatch(md, n_ranks=n_ranks, rank=r, rank_of_surface=ranks, no_fusion=(fusion == "stream"), use_graph=True) as b:
            assert b.n_surfaces_in_batch == int((ranks == r).sum())
            assert b.n_shared_zones == 0
            b.upload_state(st)
            b.march(got, w[:4], a0, b0)        # the caller-owned state: each rank writes its own slots only
            b.march_resident(w[4:], a0, b0)
            b.synchronize()
            b.download_state(got)
            total_iters += b.nomass_iterations()
    assert total_iters == iters
    assert_state_close(md, ref, got)


@pytest.mark.parametrize("rooms,n,kw", [(8, 20, {}), (8, 20, dict(no_fusion=True)), (8, 32, {}), (6, 9, {}),
                                        (40, 20, {}), (40, 32, {}), (24, 13, dict(fuse_always=True)), (100, 16, {}),
                                        (40, 7, dict(fuse_always=True, nodes_per_lane=4)),
                                        (16, 12, dict(fuse_always=True))])
def test_buildings_of_small_rooms_joined_by_partitions(oracle, rooms, n, kw):
    """Buildings as models have them (src/model.rs:556-590: a partition is in the balance of both rooms it separates):
    rooms of a dozen walls, two of them interior partitions to the next room. A building is one cluster of `rooms`
    zones: cluster-resident, its workgroup balances the zones side by side in rows of 16 lanes (more zones than
    wavefronts); 24 or 40 rooms do not fit a workgroup and are marched by a TEAM of workgroups that exchange the
    partial sums of the zones they share once per sub-timestep (layout.hpp, FusedSuper); streamed — by choice, or
    because 100 rooms do not fit a team either — k_zones gives every zone a row of 16 lanes instead of a wavefront
    (few walls per zone)."""
    per = rooms * 12
    md, st = mdl.partitioned_buildings(5 * per, n, rooms=rooms, dt=45.0, seed=rooms + n)
    Z = md["n_zones"]
    rng = np.random.default_rng(rooms)
    w = mdl.weather_series(13, 45.0, wind_speed=3.5, wind_deg=120.0)
    a0 = rng.uniform(0., 60., Z)
    b0 = rng.uniform(0.1, 2., Z)
    ref, got, iters, gpu_iters, counts = run_both(oracle, md, st, w, a0, b0, **kw)
    assert iters == gpu_iters
    assert_state_close(md, ref, got)
    with HeatBatch(md, **kw) as b:
        fits = rooms <= 40 and "no_fusion" not in kw
        assert (b.n_fused_surfaces == md["n_surfaces"]) == fits, (b.n_fused_surfaces, b.class_counts())


def test_teams_beside_short_calls_and_zone_lists_rebuilt(oracle):
    """Clusters marched by teams of workgroups (layout.hpp, FusedSuper) in a batch large enough for the planner's own
    rules (> 8192 surfaces): a call of one sub-timestep streams the teams' mixed tiles (a resident launch pays from two
    sub-timesteps on, batch.hip kFusedMinSubsteps), the next calls — one of two sub-timesteps, ThermalModel::march with
    dt_subdivisions = 2 is the reference's validation setup — march them resident; the tile lists survive
    heat_batch_set_shared_zones (a sharded host's call) with no zone shared, and sharing a team's zone is refused."""
    md, st = mdl.partitioned_buildings(9600, 20, rooms=40, dt=45.0, seed=77)
    Z = md["n_zones"]
    rng = np.random.default_rng(3)
    w = mdl.weather_series(15, 45.0, wind_speed=2.5, wind_deg=75.0)
    a0 = rng.uniform(0., 60., Z)
    b0 = rng.uniform(0.1, 2., Z)
    ref = st.copy()
    rc, iters = oracle.OracleModel(md).march(ref, w, a0, b0)
    assert rc == 0
    got = st.copy()
    with HeatBatch(md, use_graph=True) as b:
        assert b.n_fused_surfaces == md["n_surfaces"]
        b.upload_state(got)
        b.march_resident(w[:1], a0, b0)               # streamed: a single sub-timestep
        assert b.n_fused_launches == 0
        b.march_resident(w[1:3], a0, b0)              # teams, two sub-timesteps
        n2 = b.n_fused_launches
        assert n2 > 0
        b.march_resident(w[3:9], a0, b0)              # teams (the other direction through the clusters)
        assert b.n_fused_launches > n2
        b.set_shared_zones(np.zeros(0, dtype=np.int32))   # nothing shared: the lists are rebuilt, teams stay
        assert b.n_fused_surfaces == md["n_surfaces"]
        b.march_resident(w[9:], a0, b0)
        b.synchronize()
        b.download_state(got)
        with pytest.raises(HeatError):
            b.set_shared_zones(np.array([5], dtype=np.int32))
    assert_state_close(md, ref, got)


@pytest.mark.parametrize("seed", [60009, 60029, 60057])
def test_drop_in_calls_with_output_masks_the_fuzzer_found(seed):
    """tools/fuzz.py, round 3: heat_batch_march_ex on a caller-owned state, new irradiances between the calls, a random
    output mask per call. A call that left HEAT_OUT_ZONE_TEMPERATURES out left the caller's zone slots a call behind the
    device's — and the next call's upload of "what other modules write" took them and set the zones back (4e-2 K, other pass
    counts). The zone slots are now read only while the caller's state holds what the path last computed for them."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz
    line = fuzz.run_dropin_case(seed)
    assert line is not None and "drop-in" in line
    print(line)


@pytest.mark.parametrize("seed", [10802, 10846, 10968, 11073])
def test_cases_the_fuzzer_found(seed):
    """tools/fuzz.py, round 3: buildings of seven-node walls (one lane per wall, 64 walls per wavefront) rewired into one
    cluster — a team whose members hold more than sixteen zones each. A member that published and awaited its zones one
    after the other could wait for a sum its partner publishes only behind a zone the partner is itself waiting at (the march
    ended with HEAT_E_DEVICE after the bounded wait); members now publish everything before they wait for anything."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz
    line = fuzz.run_case(seed)
    assert line is not None and "fused" in line
    print(line)


@pytest.mark.parametrize("mode", ["planned", "streamed", "general"])
def test_numerical_failure_names_the_surface(mode):
    """The reference's panic on a NaN convection coefficient names the values (surface.rs:704-707); the library
    reports the first offending surface by its number in the caller's descriptor — whichever kernel it ran in."""
    md, st = mdl.clustered_massive(700, Z=28, dt=45.0, seed=3)
    bad = [411, 97, 605]                                  # 97 is the first in the caller's order
    n = np.diff(md["node_offset"])
    for s_ in bad:
        st[md["first_node_slot"][s_]] = np.nan            # NaN front temperature -> NaN hs on that surface
    kw = dict(planned={}, streamed=dict(no_fusion=True), general=dict(force_general=True))[mode]
    with HeatBatch(md, **kw) as b:
        assert b.failed_surface() == (-1, 0)
        b.upload_state(st)
        b.march_resident(mdl.weather_series(1, 45.0))
        with pytest.raises(HeatError) as e:
            b.synchronize()
        assert e.value.code > 0
        idx, kind = b.failed_surface()
        assert idx in bad and kind == e.value.code, (idx, kind, str(e.value))
        assert "surface %d" % idx in str(e.value)
        if mode != "planned":
            # one sub-timestep, streamed: only the three surfaces themselves are bad, and the smallest DEVICE number
            # wins — every reported index must be one of them; after the report the record is cleared on the device
            assert idx in bad
        b.upload_state(mdl.initial_state(md))
        b.march_resident(mdl.weather_series(2, 45.0))
        b.synchronize()                                   # healthy again


@pytest.mark.parametrize("case", ["massive_no_ir_no_solar", "mixed_no_ir_no_solar", "nomass_no_ir_no_solar", "massive_full",
                                  "mixed_full", "nomass_full", "massive_no_ir_yes_solar", "mixed_yes_ir_no_solar"])
def test_config1_energyplus_series_through_the_abi(oracle, case):
    """BASELINE config 1 (and its mixed / no-mass / full-radiation siblings): the reference's validation
    harness (validate_wall_heat_transfer.rs:615-711) driven through heat_batch_march, step by step with
    the caller-owned state, against the same harness on the oracle."""
    import os
    from test_energyplus_series import CASES, GOLD, march_series, single_zone_model
    layers, emis, sol = CASES[case]
    fx = dict(np.load(os.path.join(GOLD, "wall_%s.npz" % case)))
    fx = {k: v[:400] for k, v in fx.items()}
    md, st, d, n_sub = single_zone_model(oracle, layers, emis, sol)
    ref_state = st.copy()
    ref = march_series(oracle, md, ref_state, n_sub, fx, emis)
    got_state = st.copy()
    with HeatBatch(md) as b:
        b.upload_state(got_state)
        got = march_series(oracle, md, got_state, n_sub, fx, emis, march=lambda s, w: b.march(s, w))
    assert np.allclose(got, ref, rtol=RTOL, atol=ATOL)
    # the harness derives the IR irradiance it feeds from the front temperature of the run itself
    # (validate_wall_heat_transfer.rs:654-660): an input slot, equal between the two runs to the parity tolerance only
    ir = md["ir_front_slot"]
    assert np.allclose(got_state[ir], ref_state[ir], rtol=RTOL, atol=0.0)
    got_state[ir] = ref_state[ir]
    assert_state_close(md, ref_state, got_state)


def test_sharded_march_single_rank_nccl():
    """The multi-GPU driver (heat_amd/sharded.py) with world_size = 1 over RCCL: the same split-phase
    sequence and exchange the N-GPU bench runs, checked against the oracle. Runs in a fresh process
    because torch must be imported before the HIP library is loaded (two HIP runtimes in one process
    otherwise: torch ships its own libamdhip64)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "sharded_nccl_worker.py")], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "SHARDED OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_sharded_fused_launch_on_the_work_queue():
    """Beside an exchange loop the fused launch holds fewer workgroups than it has FusedBlocks and hands them out
    through a work queue (wavefronts without a tile keep step with the barriers). A full-size sharded batch does
    that with 496 workgroups for 10 000 blocks; here the room is shrunk to 20 slots (HEAT_AMD_FUSED_ROOM) so that
    a handful of workgroups march some 50 blocks, and the result is held to the oracle."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HEAT_AMD_FUSED_ROOM="20", HEAT_AMD_TRACE="1")
    r = subprocess.run([sys.executable, os.path.join(here, "sharded_nccl_worker.py"), "queue"], capture_output=True,
                       text=True, timeout=900, env=env)
    assert r.returncode == 0 and "SHARDED OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "fused launch on the work queue:" in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("n,npl,which", [(3, 4, "both"), (5, 4, "front"), (9, 8, "back"), (9, 4, "both"), (17, 8, "both"),
                                          (8, 8, "both"), (33, 16, "back"), (2, 4, "front"), (2, 4, "back"), (40, 8, "both")])
def test_no_mass_facings_on_the_fast_path(oracle, n, npl, which):
    """Walls whose face node(s) carry no mass (one-node no-mass chunks, surface.rs:790-898), for node counts
    that put the facing alone in a lane, at a lane boundary, or in a single-lane surface."""
    md, st = mdl.uniform_massive(333, n, Z=4, dt=45.0, seed=n + npl)
    off = md["node_offset"]
    mass = md["mass"].copy()
    u = md["uvalue"].copy()
    rng = np.random.default_rng(n)
    if which in ("front", "both"):
        mass[off[:-1]] = 0.0
        u[off[:-1]] = rng.uniform(0.5, 3.0, 333)
    if which in ("back", "both"):
        mass[off[1:] - 1] = 0.0
        u[off[1:] - 2] = rng.uniform(0.5, 3.0, 333)
    md["mass"], md["uvalue"] = mass, u
    md["front_emissivity"] = md["front_emissivity"] * 0.2
    md["back_emissivity"] = md["back_emissivity"] * 0.2
    w = mdl.weather_series(20, 45.0)
    for no_palette in (False, True):
        ref, got, iters, gpu_iters, counts = run_both(oracle, md, st, w, nodes_per_lane=npl, no_palette=no_palette)
        assert counts[3] + counts[4] == 0, counts  # all on the fast path
        assert iters == gpu_iters and iters > 0
        assert_state_close(md, ref, got)


@pytest.mark.parametrize("npl", [0, 4, 8])
def test_many_layer_walls_fall_back_to_per_node_constants(oracle, npl):
    """Walls with more distinct dt/mass or U values than the wide palette holds (13 / 7, layout.hpp) keep per-node
    arrays; walls within the limit use the palette; both in one batch, bitwise equal to the no-palette build."""
    md, st = mdl.uniform_massive(400, 24, Z=4, dt=45.0, seed=77)
    off = md["node_offset"]
    rng = np.random.default_rng(5)
    mass = md["mass"].copy()
    u = md["uvalue"].copy()
    for s_ in range(0, 400, 2):  # every other wall: 8 layers of 3 nodes with their own material
        o = off[s_]
        for layer in range(8):
            f = rng.uniform(0.7, 1.3)
            mass[o + 3 * layer:o + 3 * layer + 3] *= f
            u[o + 3 * layer:o + 3 * layer + 3] *= rng.uniform(0.7, 1.3)
        u[off[s_ + 1] - 1] = 0.0
    md["mass"], md["uvalue"] = mass, u
    w = mdl.weather_series(15, 45.0)
    ref, got, _, _, counts = run_both(oracle, md, st, w, nodes_per_lane=npl)
    assert counts[3] + counts[4] == 0
    assert_state_close(md, ref, got)
    ref2, got2, *_ = run_both(oracle, md, st, w, nodes_per_lane=npl, no_palette=True)
    assert np.array_equal(got, got2)


def test_config1_end_to_end_through_the_product_setup_and_kernels(oracle):
    """Constructions in, EnergyPlus-validated zone temperatures out, with no oracle code on the product side:
    heat_model_builder_* (ThermalModel::new) -> heat_batch_create -> heat_batch_march per caller timestep."""
    import os
    from heat_amd import ModelBuilder
    from test_energyplus_series import GOLD, march_series
    mb = ModelBuilder(20)
    z = mb.add_zone(600.)
    poly = dict(thickness=0.02, k=0.0252, rho=17.5, cp=2400., front_thermal_abs=0., back_thermal_abs=0.,
                front_solar_abs=0., back_solar_abs=0.)
    conc = dict(thickness=0.2, k=0.816, rho=1700., cp=800., front_thermal_abs=0., back_thermal_abs=0.,
                front_solar_abs=0., back_solar_abs=0.)
    mb.add_surface([poly, conc, poly], 60., 46., (0., -1., 0.), 1.5, mdl.OUTDOOR, mdl.SPACE, back_zone=z)
    md, st, n_sub = mb.finish()
    fx = dict(np.load(os.path.join(GOLD, "wall_mixed_no_ir_no_solar.npz")))
    exp = fx["zone_t"]
    ref_state = st.copy()
    ref = march_series(oracle, md, ref_state, n_sub, fx, 0.0)
    got_state = st.copy()
    with HeatBatch(md) as b:
        b.upload_state(got_state)
        got = march_series(oracle, md, got_state, n_sub, fx, 0.0, march=lambda s, w: b.march(s, w))
    assert np.allclose(got, ref, rtol=RTOL, atol=ATOL)
    rmse = float(np.sqrt(np.mean((got[5001:] - exp[5001:]) ** 2)))
    assert rmse < 0.1, rmse  # vs EnergyPlus, after the 5000-step warm-up of the reference's harness
    mb.close()


def test_cavities_on_the_fast_path_and_their_fallbacks(oracle):
    """Gas cavities between massive nodes (one or two per wall, at lane boundaries or inside a lane) take the
    fast path; a cavity next to a no-mass node or a third cavity sends the wall to the catch-all kernel."""
    rng = np.random.default_rng(3)
    S = 240
    md, st = mdl.uniform_massive(S, 24, Z=3, dt=45.0, seed=9)
    off = md["node_offset"]
    segc = np.full(off[-1], -1, dtype=np.int32)
    cav = np.zeros(2 * S + 40, dtype=mdl.CAVITY_DTYPE)
    mass = md["mass"].copy()
    nc = 0
    for s_ in range(S):
        o = off[s_]
        spots = [(7,), (15,), (3, 16), (7, 8), (11, 19), (0,), (22,), (5, 10, 15)][s_ % 8]
        for p_ in spots:
            segc[o + p_] = nc
            cav[nc] = (rng.uniform(0.01, 0.05), 1.0, rng.choice([math.pi / 2, 1.0, 2.3, 0.3]), 0.84, rng.uniform(0.1, 0.9),
                       rng.integers(0, 4), 0)
            nc += 1
        if s_ % 16 == 9:
            mass[o] = 0.0  # a no-mass facing next to ... nothing special: cavity at 7/15 is far from it
        if s_ % 16 == 5:
            mass[o] = 0.0  # cavity at segment 0 touches the no-mass node 0 -> catch-all
    md["seg_cavity"], md["cavities"], md["mass"] = segc, cav[:nc], mass
    md["front_emissivity"] = md["front_emissivity"] * 0.2
    w = mdl.weather_series(20, 45.0)
    for npl in (0, 4, 8, 16):
        ref, got, iters, gpu_iters, counts = run_both(oracle, md, st, w, nodes_per_lane=npl)
        assert sum(counts[:3]) > 0 and counts[4] > 0, counts
        assert iters == gpu_iters
        assert_state_close(md, ref, got)


def test_edge_cases_empty_tiny_and_long_walls(oracle):
    """Empty batch, zones without surfaces, one-node and very long walls (catch-all kernel), NaN / negative solar
    input (clamps of surface.rs:916-923), zero wind, all through the C ABI."""
    # empty batch with zones: marching only applies the zone formula with a0/b0
    md = mdl.empty(0, 2, 45.0)
    md["node_offset"] = np.zeros(1, dtype=np.int64)
    for k in ("mass", "uvalue", "front_alpha", "back_alpha", "front_ambient", "back_ambient", "front_emissivity",
              "back_emissivity", "area", "perimeter", "cos_tilt", "normal_x", "normal_y", "wind_modifier"):
        md[k] = np.zeros(0)
    for k in ("front_kind", "back_kind", "front_zone", "back_zone"):
        md[k] = np.zeros(0, dtype=np.int32)
    md["zone_volume"] = np.array([50., 80.])
    st = mdl.layout_state(md)
    w = mdl.weather_series(5, 45.0)
    a0, b0 = np.array([300., 0.]), np.array([10., 0.])
    ref, got, *_ = run_both(oracle, md, st, w, a0, b0)
    assert np.allclose(got, ref, rtol=RTOL, atol=ATOL) and got[1] == 22.0 and got[0] != 22.0

    # long walls (n = 200 > 64 lanes * ... still fast path) and n = 1000 (k > 64 at M = 4 -> catch-all when forced)
    for n, npl in ((200, 4), (1000, 4), (1000, 16)):
        md, st = mdl.uniform_massive(9, n, Z=2, dt=20.0, seed=n)
        ref, got, _, _, counts = run_both(oracle, md, st, mdl.weather_series(6, 20.0), nodes_per_lane=npl)
        assert_state_close(md, ref, got)
        if n == 1000 and npl == 4:
            assert counts[4] == 9  # 250 lanes per surface do not fit a wavefront

    # weird inputs
    md, st = mdl.ragged_mixed(300, Z=3, dt=45.0, seed=4)
    st[md["solar_front_slot"][::3]] = np.nan
    st[md["solar_front_slot"][1::3]] = -50.0
    st[md["solar_back_slot"][::2]] = np.nan
    st[md["solar_back_slot"][1::2]] = -20.0   # passes through unclamped (the reference's quirk)
    w = mdl.weather_series(8, 45.0, wind_speed=0.0)
    ref, got, iters, gpu_iters, _ = run_both(oracle, md, st, w)
    assert iters == gpu_iters
    assert_state_close(md, ref, got)


def test_one_node_surfaces(oracle):
    """A single-node wall (n = 1: both faces on the same node), massive and no-mass."""
    for mass in (5.0e4, 0.0):
        segs = dict(mass=np.array([mass]), uvalue=np.array([0.0]), front_alpha=np.array([0.3]),
                    back_alpha=np.array([0.2]))
        md, st = surfaces_model(segs, 30.0, mdl.OUTDOOR, mdl.SPACE, n_zones=1, zone_volume=[30.], front_emis=0.1,
                                back_emis=0.1, cos_tilt=0.0, normal=(0., -1., 0.), copies=70)
        st[md["solar_front_slot"]] = 200.0
        st[md["ir_front_slot"]] = mdl.SIGMA * 283.15 ** 4
        st[mdl.node_slots(md)] = np.linspace(15., 30., 70)
        ref, got, iters, gpu_iters, counts = run_both(oracle, md, st, mdl.weather_series(10, 30.0))
        assert iters == gpu_iters
        assert_state_close(md, ref, got)


@pytest.mark.parametrize("case", ["nomass_wallonly", "heater_on", "heater_and_infiltration"])
def test_closed_form_cases_through_the_abi(oracle, case):
    """The reference's closed-form validation cases (validate_wall_heat_transfer.rs:181-613) with the host-side
    zone terms a0 / b0 (heater, infiltration) handed to heat_batch_march, against the oracle and the closed form."""
    from test_energyplus_series import CLOSED_FORM, closed_form_case
    n, steps, kw = CLOSED_FORM[case]
    steps = min(steps, 300)
    md, ref_state, ref, exp = closed_form_case(oracle, n, steps, **kw)
    holder = {}

    def gpu_march(state, w, a0, b0):
        if "b" not in holder:
            holder["b"] = HeatBatch(md)
            holder["b"].upload_state(state)
        holder["b"].march(state, w, a0, b0)

    md2, got_state, got, _ = closed_form_case(oracle, n, steps, march=gpu_march, **kw)
    holder["b"].close()
    assert np.allclose(got, ref, rtol=RTOL, atol=ATOL)
    assert_state_close(md, ref_state, got_state)
    assert np.abs(got - exp).max() < 0.35


def test_compiled_cpp_host_on_the_c_abi(oracle):
    """examples/march_walls.cpp (no Python in the loop: model builder -> heat_batch_create -> heat_batch_march on a
    caller-owned state) prints what the oracle computes for the same building."""
    import subprocess
    from heat_amd import ModelBuilder, build as hb
    exe = hb.build_example()
    n_walls, n_steps = 12, 10
    out = subprocess.run([exe, str(n_walls), str(n_steps)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    # the same building, built here
    ins = dict(thickness=0.02, k=0.0252, rho=17.5, cp=2400., front_thermal_abs=0.2, back_thermal_abs=0.2,
               front_solar_abs=0.7, back_solar_abs=0.7)
    conc = dict(thickness=0.2, k=0.816, rho=1700., cp=800., front_thermal_abs=0.9, back_thermal_abs=0.9,
                front_solar_abs=0.7, back_solar_abs=0.7)
    mb = ModelBuilder(20)
    za, zb = mb.add_zone(600.0), mb.add_zone(250.0)
    for i in range(n_walls):
        layers = [conc] if i % 3 == 0 else ([ins, conc, ins] if i % 3 == 1 else [ins])
        area = 10.0 + i
        az = 0.4 * i
        mb.add_surface(layers, area, 2.0 * (area / 3.0 + 3.0), (math.sin(az), math.cos(az), 0.0), 1.5 + 3.0 * (i % 4),
                       mdl.SPACE if i % 5 == 4 else mdl.OUTDOOR, mdl.SPACE, front_zone=zb, back_zone=zb if i % 2 else za)
    md, state, n_sub = mb.finish()
    m = oracle.OracleModel(md)
    for step in range(n_steps):
        s_ = np.arange(n_walls)
        state[md["solar_front_slot"]] = 50.0 * (step % 7) + 3.0 * s_
        state[md["ir_front_slot"]] = 5.670374419e-8 * (283.15 + step) ** 4
        state[md["ir_back_slot"]] = 5.670374419e-8 * 295.15 ** 4
        w = np.tile([10.0 + 0.5 * step, (150.0 + 10.0 * step) * (math.pi / 180.0), 2.0 + 0.1 * step], (n_sub, 1))
        rc, _ = m.march(state, w, np.array([150.0, 0.0]), np.array([0.0, 0.0]))
        assert rc == 0
    head = lines[0].split()
    assert int(head[1]) == md["n_state"] and float(head[3]) == md["dt"] and int(head[5]) == n_sub
    zones = [float(l.split()[2]) for l in lines if l.startswith("zone")]
    assert np.allclose(zones, state[md["zone_slot"]], rtol=RTOL, atol=ATOL)
    surf = np.array([[float(x) for x in l.split()[2:]] for l in lines if l.startswith("surface")])
    n = np.diff(md["node_offset"])
    exp = np.stack([state[md["first_node_slot"]], state[md["first_node_slot"] + n - 1], state[md["hs_front_slot"]],
                    state[md["flow_back_slot"]]], axis=1)
    assert np.allclose(surf, exp, rtol=RTOL, atol=ATOL)
    mb.close()


@pytest.mark.parametrize("npl", [0, 4, 8])
def test_cluster_resident_march_matches_the_oracle_and_the_streamed_march(oracle, npl):
    """Zone-connected clusters that fit a workgroup march all sub-timesteps of a call in one launch (temperatures in
    registers, zone balance in LDS); the others are streamed beside them. Same results as the oracle (1e-9) and as
    the all-streamed march (no_fusion), also when the march is cut into several calls."""
    md, st = mdl.clustered_massive(1500, Z=60, dt=45.0, seed=11 + npl)
    # the first forty zones are chained by some of their walls into one cluster of ~1000 walls: too large for a
    # workgroup and for a team of eight, it is streamed beside the others
    chain = (md["back_zone"] < 40) & (np.arange(1500) % 3 == 0)
    md["front_kind"] = np.where(chain, mdl.SPACE, md["front_kind"]).astype(np.int32)
    md["front_zone"] = np.where(chain, (md["back_zone"] + 1) % 40, md["front_zone"]).astype(np.int32)
    w = mdl.weather_series(23, 45.0, wind_speed=3.5, wind_deg=120.0)
    a0 = np.linspace(0., 60., 60)
    b0 = np.linspace(0., 2., 60)
    ref = st.copy()
    rc, iters = oracle.OracleModel(md).march(ref, w, a0, b0)
    assert rc == 0
    got = st.copy()
    with HeatBatch(md, nodes_per_lane=npl, fuse_always=True) as b:
        nf = b.n_fused_surfaces
        assert 0 < nf < md["n_surfaces"]          # both kinds of clusters are present
        b.upload_state(got)
        b.march(got, w[:9], a0, b0)               # one launch of 9 sub-timesteps ...
        b.march(got, w[9:10], a0, b0)             # ... of one ...
        b.march(got, w[10:], a0, b0)              # ... of 13
        assert b.nomass_iterations() == iters
        assert b.n_fused_launches >= 3
    assert_state_close(md, ref, got)
    streamed = st.copy()
    with HeatBatch(md, nodes_per_lane=npl, no_fusion=True) as b:
        assert b.n_fused_surfaces == 0
        b.upload_state(streamed)
        b.march(streamed, w, a0, b0)
    assert np.allclose(got, streamed, rtol=1e-11, atol=1e-11)
    # switched off at run time: the same layout, everything streamed
    off = st.copy()
    with HeatBatch(md, nodes_per_lane=npl, use_graph=True, fuse_always=True) as b:
        b.set_fusion(False)
        b.upload_state(off)
        b.march(off, w, a0, b0)
    assert np.allclose(got, off, rtol=1e-11, atol=1e-11)


def test_cluster_resident_march_config2_and_lone_surfaces(oracle):
    # BASELINE config 2 (identical massive walls, 100 per zone): every cluster fused; plus walls that face no zone
    md, st = mdl.uniform_massive(400, 20, Z=4, dt=90.0, identical=True, vertical=True)
    w = mdl.weather_series(50, 90.0)
    ref, got, _, _, counts = run_both(oracle, md, st, w)
    assert_state_close(md, ref, got)
    with HeatBatch(md) as b:
        assert b.n_fused_surfaces == 400
    md, st = mdl.uniform_massive(300, 24, Z=3, dt=45.0, seed=5)
    md["back_kind"][:] = mdl.OUTDOOR      # Outdoor on both sides: coupled to no zone
    md["front_kind"][::3] = mdl.AMBIENT
    md["front_ambient"][::3] = 17.5
    w = mdl.weather_series(30, 45.0)
    ref, got, _, _, counts = run_both(oracle, md, st, w, fuse_always=True)
    assert_state_close(md, ref, got)
    with HeatBatch(md, fuse_always=True) as b:
        assert b.n_fused_surfaces == 300


def test_walls_of_many_materials_keep_the_palette_and_the_resident_march(oracle):
    """A wall of five or six materials (render / brick / insulation / block / plaster) has more distinct conductances
    than the narrow palette holds; the batch then stores 16-double palettes (layout.hpp kPal) and the wall stays in
    the register kernel and in the fused clusters. Narrow and wide batches give the same numbers for narrow walls."""
    mats = [dict(thickness=0.02, k=0.9, rho=1800., cp=840.), dict(thickness=0.11, k=0.6, rho=1600., cp=840.),
            dict(thickness=0.08, k=0.25, rho=600., cp=1000.), dict(thickness=0.14, k=0.51, rho=1400., cp=1000.),
            dict(thickness=0.06, k=1.4, rho=2100., cp=880.), dict(thickness=0.015, k=0.4, rho=1000., cp=1000.)]
    mats = [dict(L, front_thermal_abs=0.9, back_thermal_abs=0.9, front_solar_abs=0.6, back_solar_abs=0.6) for L in mats]
    # (three materials: more distinct V = dt / C than the TINY palette of 4 + 4 entries holds, the batch stores the
    # narrow 8 + 4; one or two materials — most other tests — take the tiny one)
    for nl, copies in ((3, 96), (5, 96), (6, 64)):
        d = oracle.discretize(mats[:nl], 600., 0.03, 60., 1., math.pi / 2)
        if nl >= 5:
            assert len(set(np.round(d["uvalue"], 12))) > 4, d["uvalue"]   # (entry 0 of the palette is 0.0)
        else:
            nv = len(set(np.round(600. / d["tstep_subdivision"] / d["mass"][d["mass"] >= 1e-5], 9)))
            assert 4 <= nv <= 7 and len(set(np.round(d["uvalue"], 12))) <= 4, (nv, d["uvalue"])
        assert not oracle.get_chunks(d["mass"])[1]                         # all massive
        dt = 600. / d["tstep_subdivision"]
        md, st = surfaces_model(d, dt, mdl.OUTDOOR, mdl.SPACE, n_zones=4, zone_volume=[60.] * 4, front_emis=0.9,
                                back_emis=0.9, area=10., perimeter=13., cos_tilt=0.0, normal=(0., 1., 0.), copies=copies)
        md["back_zone"] = (np.arange(copies) % 4).astype(np.int32)
        rng = np.random.default_rng(nl)
        mdl.perturb_initial_temperatures(md, st, rng)
        st[md["solar_front_slot"]] = rng.uniform(0., 500., copies)
        mdl.set_ir_from_air(md, st, 5.0)
        w = mdl.weather_series(24, dt, wind_speed=3.0, wind_deg=200.0)
        ref, got, _, _, counts = run_both(oracle, md, st, w, np.full(4, 30.), np.full(4, 1.0))
        assert counts[3] == 0 and counts[4] == 0, counts                   # nothing in the one-lane kernels
        assert_state_close(md, ref, got)
        with HeatBatch(md) as b:
            assert b.n_fused_surfaces == copies
        streamed = st.copy()
        with HeatBatch(md, no_fusion=True) as b:
            b.upload_state(streamed)
            b.march(streamed, w, np.full(4, 30.), np.full(4, 1.0))
        assert np.allclose(got, streamed, rtol=1e-10, atol=1e-10)


def test_cluster_resident_march_with_gas_cavities(oracle):
    # Trombe-like walls (massive / gas cavity / massive) in fused clusters: Cavity::u_value once per sub-timestep,
    # inside the resident loop (4 or 8 nodes per lane)
    md, st = mdl.glazing_cavity(480, Z=8, dt=45.0, seed=13, trombe_fraction=1.0)
    w = mdl.weather_series(25, 45.0)
    ref, got, iters, gpu_iters, counts = run_both(oracle, md, st, w, fuse_always=True)
    assert iters == gpu_iters
    assert_state_close(md, ref, got)
    with HeatBatch(md, fuse_always=True) as b:
        assert b.n_fused_surfaces == 480, (b.n_fused_surfaces, b.class_counts())
        b.upload_state(st.copy())
        b.march_resident(w[:4])
        b.synchronize()
        assert b.n_fused_launches > 0
    streamed = st.copy()
    with HeatBatch(md, no_fusion=True) as b:
        b.upload_state(streamed)
        b.march(streamed, w)
    assert np.allclose(got, streamed, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("npl", [0, 4, 8, 16])
def test_cluster_resident_march_of_rooms_with_windows(oracle, npl):
    """Mixed workgroups: the walls of a room on fast-path wavefronts, its double-glazed windows and thin no-mass
    partitions on small-surface wavefronts of the same workgroup (one lane per surface, small_step), one zone balance.
    (fuse_always: the planner's cost model would stream rooms with double glazing — the machinery is tested here.)"""
    md, st = mdl.rooms_with_windows(1400, Z=70, dt=45.0, seed=23 + npl)
    w = mdl.weather_series(17, 45.0, wind_speed=2.5, wind_deg=75.0)
    a0 = np.linspace(0., 80., 70)
    b0 = np.linspace(0., 1.5, 70)
    ref = st.copy()
    rc, iters = oracle.OracleModel(md).march(ref, w, a0, b0)
    assert rc == 0
    got = st.copy()
    with HeatBatch(md, nodes_per_lane=npl, fuse_always=True) as b:
        counts = b.class_counts()
        assert counts[3] > 0                       # windows / thin partitions are there ...
        # ... and most rooms march cluster-resident, windows included (a forced blocking factor of 8 or 16 nodes per
        # lane leaves the rooms with short walls — one lane per wall — to the streamed kernels)
        assert b.n_fused_surfaces > (700 if npl in (0, 4) else -1)
        b.upload_state(got)
        b.march(got, w[:8], a0, b0)
        b.march(got, w[8:], a0, b0)
        assert b.nomass_iterations() == iters
        assert (b.n_fused_launches > 0) == (b.n_fused_surfaces > 0)
    assert_state_close(md, ref, got)
    streamed = st.copy()
    with HeatBatch(md, nodes_per_lane=npl, no_fusion=True) as b:
        b.upload_state(streamed)
        b.march(streamed, w, a0, b0)
    assert np.allclose(got, streamed, rtol=1e-10, atol=1e-10)


def test_headline_at_full_size(oracle):
    """BASELINE's headline size (1 000 000 walls x 32 nodes, 10 000 zones) through properties that do not need the
    oracle at that size: the cluster-resident march and the streamed march — different kernels, different data
    paths — agree to 1e-10 on every owned slot after 23 sub-timesteps; 300 walls (three whole zones, so their zone
    balance is complete) are checked against the oracle marching that sub-model alone; a second batch gives the same
    bits (run-to-run determinism)."""
    S, n, Z = 1_000_000, 32, 10_000
    md, st = mdl.uniform_massive(S, n, Z=Z, dt=45.0)
    w = mdl.weather_series(23, 45.0)
    outs = []
    for kw in (dict(), dict(no_fusion=True), dict()):
        got = st.copy()
        with HeatBatch(md, **kw) as b:
            assert (b.n_fused_surfaces == S) == ("no_fusion" not in kw)
            b.upload_state(got)
            b.march_resident(w[:20])
            b.march_resident(w[20:])
            b.synchronize()
            b.download_state(got)
        outs.append(got)
    assert np.all(np.isfinite(outs[0]))
    assert np.allclose(outs[0], outs[1], rtol=1e-10, atol=1e-10)
    assert np.array_equal(outs[0], outs[2])
    # three whole zones against the oracle (uniform_massive: 100 consecutive walls per zone, front Outdoor / back Space)
    idx = np.arange(4200, 4500)
    assert set(md["back_zone"][idx]) == {42, 43, 44} and np.all(md["front_kind"][idx] == mdl.OUTDOOR)
    sub = mdl.subset(md, idx)
    ref = st.copy()
    rc, _ = oracle.OracleModel(sub).march(ref, w)
    assert rc == 0
    for name, sl in (("nodes", mdl.node_slots(sub)), ("hs", sub["hs_back_slot"]), ("flow", sub["flow_front_slot"]),
                     ("zones", md["zone_slot"][42:45])):
        assert np.allclose(outs[0][sl], ref[sl], rtol=RTOL, atol=ATOL), name


def test_config3_at_full_size_fast_kernels_equal_the_catch_all_kernel():
    """BASELINE config 3 at its full size (1 000 000 ragged mixed surfaces, 8-64 nodes): the lane-blocked fast kernels
    (+ the small-surface kernel) and the catch-all kernel — one lane per surface, tri-diagonal matrices in scratch, the
    reference's operation order — are independent implementations of the same path; they agree on every owned slot
    and on the number of passes of the no-mass loop."""
    md, st = mdl.ragged_mixed(1_000_000, dt=45.0)
    w = mdl.weather_series(6, 45.0)
    res = []
    for kw in (dict(), dict(force_general=True)):
        got = st.copy()
        with HeatBatch(md, **kw) as b:
            counts = b.class_counts()
            assert (counts[4] == 1_000_000) == ("force_general" in kw)
            b.upload_state(got)
            b.march_resident(w)
            b.synchronize()
            b.download_state(got)
            res.append((got, b.nomass_iterations()))
    assert res[0][1] == res[1][1] > 0
    assert np.all(np.isfinite(res[0][0]))
    assert np.allclose(res[0][0], res[1][0], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("config", ["headline", "3", "partitions"])
def test_full_size_against_the_whole_oracle(oracle, config):
    """The oracle, threaded over the surfaces on the box's host cores (the reference's disabled rayon path,
    model.rs:113-116), marches the FULL BASELINE sizes in seconds: 1 000 000 x 32 (cluster-resident march), the
    1 000 000 ragged mixed surfaces of config 3 (one streamed launch per sub-timestep) and a million walls in buildings
    of 8 rooms joined by partitions (cluster-resident, zones balanced in rows of 16 lanes) are compared slot by slot."""
    if config == "headline":
        md, st = mdl.uniform_massive(1_000_000, 32, Z=10_000, dt=45.0)
        n_sub = 10
    elif config == "partitions":
        md, st = mdl.partitioned_buildings(1_000_000, 32, dt=45.0)
        n_sub = 10
    else:
        md, st = mdl.ragged_mixed(1_000_000, dt=45.0)
        n_sub = 8
    w = mdl.weather_series(n_sub, 45.0, wind_speed=3.2, wind_deg=230.0)
    rng = np.random.default_rng(77)
    a0 = rng.uniform(0., 200., int(md["n_zones"]))
    b0 = rng.uniform(0., 5., int(md["n_zones"]))
    ref = st.copy()
    rc, _ = oracle.OracleModel(md).march(ref, w, a0, b0, threads=16)
    assert rc == 0
    got = st.copy()
    with HeatBatch(md, use_graph=True) as b:
        assert (b.n_fused_surfaces > 0) == (config != "3")
        b.upload_state(got)
        b.march(got, w, a0, b0)
    assert_state_close(md, ref, got)


@pytest.mark.parametrize("seed", range(6))
def test_planner_stress_random_zone_graphs(oracle, seed):
    """Random small models — walls, facings, windows, partitions, walls between random pairs of zones, zones nobody faces,
    walls facing the same zone on both sides, walls facing no zone — through every planner mode (cost model, everything
    that can be fused, nothing fused) and blocking factor: the same state as the oracle's."""
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.integers(60, 900))
    Z = int(rng.integers(3, 40))
    gen = mdl.rooms_with_windows if seed % 2 else mdl.clustered_massive
    md, st = gen(S, Z=Z, dt=45.0, seed=seed)
    # rewire some walls: random zone pairs (longer chains of zones), same zone on both sides, no zone at all
    pick = rng.random(S)
    both = (md["front_kind"] == mdl.SPACE) & (md["back_kind"] == mdl.SPACE)
    rew = both & (pick < 0.3)
    md["front_zone"] = np.where(rew, rng.integers(0, Z, S), md["front_zone"]).astype(np.int32)
    same = both & (pick > 0.9)
    md["front_zone"] = np.where(same, md["back_zone"], md["front_zone"]).astype(np.int32)
    nodes = np.diff(md["node_offset"])
    lone = (pick > 0.5) & (pick < 0.56) & (nodes > 4)
    md["front_kind"] = np.where(lone, mdl.AMBIENT, md["front_kind"]).astype(np.int32)
    md["back_kind"] = np.where(lone, mdl.OUTDOOR, md["back_kind"]).astype(np.int32)
    md["front_ambient"] = np.where(lone, 12.5, md["front_ambient"])
    w = mdl.weather_series(9, 45.0, wind_speed=float(rng.uniform(0.5, 6.0)), wind_deg=float(rng.uniform(0, 360)))
    a0 = rng.uniform(0., 50., Z)
    b0 = rng.uniform(0., 2., Z)
    ref = st.copy()
    rc, iters = oracle.OracleModel(md).march(ref, w, a0, b0)
    assert rc == 0
    npl = [0, 4, 8, 16][seed % 4]
    fused_counts = []
    for kw in (dict(), dict(fuse_always=True), dict(no_fusion=True), dict(fuse_always=True, nodes_per_lane=npl)):
        got = st.copy()
        with HeatBatch(md, **kw) as b:
            b.upload_state(got)
            b.march(got, w[:4], a0, b0)
            b.march(got, w[4:], a0, b0)
            assert b.nomass_iterations() == iters, kw
            assert (b.n_fused_launches > 0) == (b.n_fused_surfaces > 0), kw
            fused_counts.append(b.n_fused_surfaces)
        assert_state_close(md, ref, got)
    print("seed %d: S=%d Z=%d fused surfaces by mode %s" % (seed, S, Z, fused_counts))
    assert fused_counts[2] == 0 and fused_counts[1] >= fused_counts[0]


def test_long_march_stays_within_tolerance(oracle):
    """3 000 sub-timesteps (37.5 simulated hours at dt = 45 s, a full day of the weather cycle) in march calls of 100:
    the cluster-resident march, the streamed march and the oracle stay within the parity tolerance — rounding
    differences do not accumulate (the conduction operator is contractive)."""
    md, st = mdl.clustered_massive(400, Z=16, dt=45.0, seed=8)
    w = mdl.weather_series(3000, 45.0, wind_speed=3.0, wind_deg=210.0)
    ref = st.copy()
    rc, iters = oracle.OracleModel(md).march(ref, w)
    assert rc == 0
    for kw in (dict(fuse_always=True), dict(no_fusion=True)):
        got = st.copy()
        with HeatBatch(md, use_graph=True, **kw) as b:
            b.upload_state(got)
            for i in range(0, 3000, 100):
                b.march_resident(w[i:i + 100])
            b.synchronize()
            b.download_state(got)
            assert b.nomass_iterations() == iters
        assert_state_close(md, ref, got)
