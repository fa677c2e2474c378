"""The N > 1 path on CPU: world_size = 2, gloo.

Product code under test (heat_amd/sharded.py): the shard plan, the model-dict shard, and
ZoneExchange.all_gather with its [rank][2][n_zones] block layout. The per-rank surface work and the
zone formula are done by the oracle / numpy here (a GPU is needed for the real kernels; the same
sequence on the GPU is test_parity_gpu.py::test_split_phase_steps_equal_fused_march and
test_sharded_single_gpu_nccl).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from heat_amd import modeldict as mdl
    from heat_amd.sharded import ZoneExchange, agree_on_shared_zones, shard_model, shard_ranges
    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        md, state0 = mdl.ragged_mixed(600, Z=6, dt=45.0, seed=9)
        # walls between neighbouring zones make zones span both ranks
        weather = mdl.weather_series(6, 45.0)
        a0 = np.linspace(0., 100., 6)
        b0 = np.linspace(0., 5., 6)
        Z = 6

        ref = state0.copy()
        rc, _ = orc.OracleModel(md).march(ref, weather, a0, b0)
        assert rc == 0

        shard = shard_model(md, rank, world)
        bounds = shard_ranges(md["n_surfaces"], world)
        assert shard["n_surfaces"] == bounds[rank + 1] - bounds[rank]
        om = orc.OracleModel(shard)
        # zones this rank touches (what heat_batch_touched_zones reports) -> shared list, agreed over gloo
        touched = np.zeros(Z, dtype=np.uint8)
        touched[shard["front_zone"][shard["front_kind"] == mdl.SPACE]] = 1
        touched[shard["back_zone"][shard["back_kind"] == mdl.SPACE]] = 1
        shared = agree_on_shared_zones(touched, torch.device("cpu"))
        assert len(shared) > 0 and len(shared) < Z          # some zones span both ranks, some do not
        ns = len(shared)
        ex = ZoneExchange(ns, torch.device("cpu"))
        assert ex.world == world
        state = state0.copy()
        mine = np.nonzero(touched)[0]
        local_only = np.setdiff1d(mine, shared)
        for i in range(len(weather)):
            t_cur = state[md["zone_slot"]].copy()
            rc, _ = om.iterate_surfaces(state, weather[i, 1], weather[i, 2], weather[i, 0])
            assert rc == 0
            a, b, c = om.zones_abc(state)          # this rank's partial sums (+ c from the zone state)
            ft = t_cur.copy()
            # zones only this rank touches: finished locally (k_zones mode 2)
            at, bt = a0 + a, b0 + b
            upd = np.where(np.abs(bt) > 1e-9, at / bt + (t_cur - at / bt) * np.exp(-bt * md["dt"] / c), t_cur)
            ft[local_only] = upd[local_only]
            # shared zones: compact exchange, rank-ordered sum (k_zone_update_shared)
            ex.partial[:ns] = torch.from_numpy(a[shared])
            ex.partial[ns:2 * ns] = torch.from_numpy(b[shared])
            g = ex.all_gather().numpy().reshape(world, -1)[:, :2 * ns].reshape(world, 2, ns)
            at, bt = a0[shared].copy(), b0[shared].copy()
            for r in range(world):
                at += g[r, 0]
                bt += g[r, 1]
            cs, ts = c[shared], t_cur[shared]
            ft[shared] = np.where(np.abs(bt) > 1e-9, at / bt + (ts - at / bt) * np.exp(-bt * md["dt"] / cs), ts)
            state[md["zone_slot"]] = ft
        # the zones this rank touches equal the single-process result; shared zones are identical on both ranks
        assert np.allclose(state[md["zone_slot"]][mine], ref[md["zone_slot"]][mine], rtol=1e-9, atol=1e-9)
        zt = torch.from_numpy(state[md["zone_slot"]][shared].copy())
        both = torch.zeros(world * ns, dtype=torch.float64)
        dist.all_gather_into_tensor(both, zt)
        both = both.numpy().reshape(world, ns)
        assert np.array_equal(both[0], both[1])
        # this rank's surfaces match the single-process march
        ns = mdl.node_slots(shard)
        assert np.allclose(state[ns], ref[ns], rtol=1e-9, atol=1e-9)
        for k in ("hs_front_slot", "hs_back_slot", "flow_front_slot", "flow_back_slot"):
            assert np.allclose(state[shard[k]], ref[shard[k]], rtol=1e-9, atol=1e-9)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_two_rank_zone_exchange_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _sharded_oracle_march(md, state0, shard, touched, shared, owned, ex, weather, a0, b0, om):
    """One rank's sharded march with the oracle doing the kernels' work: the product's roles (owned / shared zones)
    and exchange layout decide what is finished where. Returns the rank's state."""
    import torch
    ns = len(shared)
    state = state0.copy()
    own_idx = np.nonzero(owned)[0]
    local_only = np.setdiff1d(own_idx, shared)
    for i in range(len(weather)):
        t_cur = state[md["zone_slot"]].copy()
        rc, _ = om.iterate_surfaces(state, weather[i, 1], weather[i, 2], weather[i, 0])
        assert rc == 0
        a, b, c = om.zones_abc(state)
        ft = t_cur.copy()
        at, bt = a0 + a, b0 + b
        upd = np.where(np.abs(bt) > 1e-9, at / bt + (t_cur - at / bt) * np.exp(-bt * md["dt"] / c), t_cur)
        ft[local_only] = upd[local_only]          # zones this rank owns alone (k_zones)
        if ns:
            ex.partial[:ns] = torch.from_numpy(a[shared])
            ex.partial[ns:2 * ns] = torch.from_numpy(b[shared])
            g = ex.all_gather().numpy().reshape(ex.world, -1)[:, :2 * ns].reshape(ex.world, 2, ns)
            at, bt = a0[shared].copy(), b0[shared].copy()
            for r in range(ex.world):              # rank order (k_zone_update_shared)
                at += g[r, 0]
                bt += g[r, 1]
            cs, ts = c[shared], t_cur[shared]
            ft[shared] = np.where(np.abs(bt) > 1e-9, at / bt + (ts - at / bt) * np.exp(-bt * md["dt"] / cs), ts)
        state[md["zone_slot"]] = ft
    return state


def _partition_worker(rank, world, port, q, case):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from heat_amd import modeldict as mdl
    from heat_amd.sharded import (ZoneExchange, agree_on_zones, partition_model, shard_by_ranks, touched_mask)
    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if case == "clusters":
            # isolated pairs of zones + two zones nobody faces (they still follow a0 / b0)
            md, state0 = mdl.clustered_massive(900, Z=36, dt=45.0, seed=5)
            Z = 36
            for z in (7, 20):   # empty them: their walls move next door (same pair -> clusters stay whole)
                for key in ("front_zone", "back_zone"):
                    md[key] = np.where(md[key] == z, z ^ 1, md[key]).astype(np.int32)
        else:
            # one ring of zones (a whole building joined by interior walls): a single oversized cluster
            md, state0 = mdl.ragged_mixed(900, Z=9, dt=45.0, seed=9)
            Z = 9
        weather = mdl.weather_series(5, 45.0)
        a0 = np.linspace(10., 100., Z)
        b0 = np.linspace(0.5, 5., Z)
        ref = state0.copy()
        rc, _ = orc.OracleModel(md).march(ref, weather, a0, b0)
        assert rc == 0

        ranks, n_shared = partition_model(md, world)      # the product's partition (host-only C ABI)
        shard = shard_by_ranks(md, ranks, rank)
        touched = touched_mask(shard)
        shared, owned = agree_on_zones(touched, torch.device("cpu"))
        assert len(shared) == n_shared, (len(shared), n_shared)
        if case == "clusters":
            assert n_shared == 0                           # cut along the clusters: nothing to exchange
            # the zones nobody faces are owned by exactly one rank each
            t = torch.from_numpy(owned[[7, 20]].astype(np.int32))
            dist.all_reduce(t)
            assert t.tolist() == [1, 1]
            assert owned[7] == (7 % world == rank) and owned[20] == (20 % world == rank)
        else:
            assert 0 < n_shared <= 2 * world
            # a zone at a cut is faced from exactly the ranks on either side of it
            cnt = torch.from_numpy(touched.astype(np.int32))
            dist.all_reduce(cnt)
            assert int(cnt.max()) <= 2 + (world > 2), cnt.tolist()
        ex = ZoneExchange(len(shared), torch.device("cpu"))
        om = orc.OracleModel(shard)
        state = _sharded_oracle_march(md, state0, shard, touched, shared, owned, ex, weather, a0, b0, om)
        own_idx = np.nonzero(owned)[0]
        assert np.allclose(state[md["zone_slot"]][own_idx], ref[md["zone_slot"]][own_idx], rtol=1e-9, atol=1e-9)
        # every zone is owned somewhere: the union over the ranks is the whole model
        cover = torch.from_numpy(owned.astype(np.int32))
        dist.all_reduce(cover)
        assert int(cover.min()) >= 1
        nsl = mdl.node_slots(shard)
        assert np.allclose(state[nsl], ref[nsl], rtol=1e-9, atol=1e-9)
        for k in ("hs_front_slot", "hs_back_slot", "flow_front_slot", "flow_back_slot"):
            assert np.allclose(state[shard[k]], ref[shard[k]], rtol=1e-9, atol=1e-9)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["clusters", "ring"])
def test_three_ranks_with_the_cluster_partition(case):
    """heat_partition over 3 ranks. Isolated clusters: no zone is shared, no collective is issued, the zones nobody
    faces are finished by rank z % 3 from a0 / b0 alone (model.rs:410-423). One oversized cluster (a ring of zones):
    cut by surface ranges, the zones at the cuts are exchanged; everything equals the single-process march."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_partition_worker, args=(r, 3, port, q, case)) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _fallback_worker(rank, world, port, q, failing_rank):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from heat_amd.sharded import ZoneExchange, native_comm_or_fallback

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = {"init": 0, "destroy": 0}

        def comm_init():           # heat_batch_comm_init as it behaved on the one-GPU rehearsal: HEAT_E_COMM on a rank
            calls["init"] += 1
            if rank == failing_rank:
                raise RuntimeError("[-7] ncclCommInitRank failed: invalid usage")

        def comm_destroy():
            calls["destroy"] += 1

        def agree_min(v):
            t = torch.tensor([v], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item())

        logged = []
        done, why = native_comm_or_fallback(comm_init, comm_destroy, agree_min, log=logged.append)
        assert calls["init"] == 1
        if failing_rank is None:
            assert done and why is None and calls["destroy"] == 0 and not logged
        else:
            # EVERY rank falls back, the ones that held a communicator have given it up, nobody raised
            assert not done and why and len(logged) == 1
            assert calls["destroy"] == (0 if rank == failing_rank else 1)
            assert ("invalid usage" in why) == (rank == failing_rank)
        # the collective the fallback uses works on every rank afterwards (nobody is stuck in, or missing from, a group)
        ex = ZoneExchange(3, torch.device("cpu"))
        ex.partial[:] = float(rank + 1)
        g = ex.all_gather().numpy().reshape(world, -1)
        assert [float(x) for x in g[:, 0]] == [float(r + 1) for r in range(world)]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("failing_rank", [1, 0, None])
def test_comm_init_failure_on_one_rank_sends_every_rank_to_the_fallback(failing_rank):
    """ShardedMarch's decision around heat_batch_comm_init (heat_amd/sharded.py, native_comm_or_fallback): a
    HEAT_E_COMM on ANY rank must neither escape (round 2: the 4-rank rehearsal ended without its JSON line) nor leave
    the ranks on different collectives."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q, failing_rank)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_zone_roles():
    from heat_amd.sharded import zone_roles
    cnt = np.array([2, 1, 0, 0, 3, 1, 0])
    local = np.array([1, 0, 0, 0, 1, 1, 0])
    shared, owned = zone_roles(cnt, local, rank=1, n_ranks=3)
    assert shared.tolist() == [0, 4]
    # faced zones, plus of the orphans (2, 3, 6) the ones with z % 3 == 1: none of 2, 3, 6 -> only faced ones... 
    assert owned.tolist() == [1, 0, 0, 0, 1, 1, 0]
    shared, owned = zone_roles(cnt, local, rank=0, n_ranks=3)
    assert owned.tolist() == [1, 0, 0, 1, 1, 1, 1]      # orphans 3 and 6 fall to rank 0


def test_shard_ranges_and_subset():
    sys.path.insert(0, ROOT)
    from heat_amd import modeldict as mdl
    from heat_amd.sharded import shard_model, shard_ranges
    b = shard_ranges(10, 4)
    assert list(b) == [0, 3, 6, 8, 10]
    md, st = mdl.ragged_mixed(50, Z=2, dt=45.0, seed=2)
    parts = [shard_model(md, r, 3) for r in range(3)]
    assert sum(p["n_surfaces"] for p in parts) == 50
    assert np.array_equal(np.concatenate([mdl.node_slots(p) for p in parts]), mdl.node_slots(md))
    assert np.array_equal(np.concatenate([p["mass"] for p in parts]), md["mass"])
    assert all(p["n_state"] == md["n_state"] and p["n_zones"] == 2 for p in parts)
