"""Worker of test_parity_gpu.py::test_sharded_march_single_rank_nccl (needs a GPU)."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from heat_amd import modeldict as mdl
from heat_amd.sharded import ShardedMarch
from oracle import oracle as orc


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    if "queue" in sys.argv:
        # HEAT_AMD_FUSED_ROOM=20 (set by the caller): the fused launches beside the exchange loop hold 4 workgroups
        # and take their FusedBlocks from the work queue (as a full-size sharded batch does with 496 of 10 000)
        assert os.environ.get("HEAT_AMD_FUSED_ROOM") == "20"
        for gen in (mdl.clustered_massive, mdl.rooms_with_windows):
            _shared_zones_beside_a_fused_march(gen, S=6000, Z=240, shared=[3, 20, 21, 100, 239])
        dist.destroy_process_group()
        print("SHARDED OK")
        return
    md, st = mdl.ragged_mixed(800, Z=8, dt=45.0, seed=21)
    w = mdl.weather_series(8, 45.0)
    ref = st.copy()
    rc, _ = orc.OracleModel(md).march(ref, w)
    assert rc == 0
    for collective in ("native", "torch"):
        sm = ShardedMarch(md, 0, 1, device_index=0, collective=collective)
        got = st.copy()
        sm.batch.upload_state(got)
        sm.march_resident(w[:3])
        sm.march_resident(w[3:])
        sm.synchronize()
        sm.batch.download_state(got)
        sm.close()
        for idx in (mdl.node_slots(md), md["hs_front_slot"], md["hs_back_slot"], md["flow_front_slot"],
                    md["flow_back_slot"], md["zone_slot"]):
            assert np.allclose(got[idx], ref[idx], rtol=1e-9, atol=1e-9)
    forced_shared_zones_single_rank("native")
    forced_shared_zones_single_rank("torch")
    shared_zones_beside_a_fused_march()
    two_shards_on_one_gpu()
    dist.destroy_process_group()
    print("SHARDED OK")


def forced_shared_zones_single_rank(collective):
    """ShardedMarch end to end on one rank with some zones declared shared, so that the exchange really runs:
    k_zones mode 2 -> all-gather of one block (native: ncclAllGather on the library's communicator inside
    heat_batch_march_resident; torch: all_gather_into_tensor between the split-phase calls) -> k_zone_update_shared."""
    md, st = mdl.ragged_mixed(3000, Z=12, dt=45.0, seed=5)
    w = mdl.weather_series(9, 45.0)
    a0 = np.linspace(0., 40., 12)
    b0 = np.linspace(0., 2., 12)
    ref = st.copy()
    rc, iters = orc.OracleModel(md).march(ref, w, a0, b0)
    assert rc == 0
    sm = ShardedMarch(md, 0, 1, device_index=0, collective=collective, force_shared=[0, 5, 11])
    assert sm.n_shared_zones == 3
    got = st.copy()
    sm.batch.upload_state(got)
    sm.march_resident(w[:4], a0, b0)
    sm.march_resident(w[4:], a0, b0)
    sm.synchronize()
    assert sm.batch.nomass_iterations() == iters
    sm.batch.download_state(got)
    # the whole-state march (upload inputs, march, download) works on a sharded batch too
    got2 = st.copy()
    sm.batch.upload_state(got2)
    if collective == "native":
        sm.batch.march(got2, w, a0, b0)
        assert np.array_equal(got2, got)
    sm.close()
    for idx in (mdl.node_slots(md), md["hs_front_slot"], md["hs_back_slot"], md["flow_front_slot"],
                md["flow_back_slot"], md["zone_slot"]):
        assert np.allclose(got[idx], ref[idx], rtol=1e-9, atol=1e-9)


def shared_zones_beside_a_fused_march():
    """Sharded batch whose clusters are fusable: the workgroups that own a shared zone are demoted to the streamed
    exchange loop (kernel -> ncclAllGather -> kernel every sub-timestep), the others march cluster-resident beside
    it on a side stream."""
    for gen in (mdl.clustered_massive, mdl.rooms_with_windows):
        _shared_zones_beside_a_fused_march(gen)


def _shared_zones_beside_a_fused_march(gen, S=1200, Z=48, shared=(3, 20, 21, 47)):
    md, st = gen(S, Z=Z, dt=45.0, seed=3)
    w = mdl.weather_series(11, 45.0)
    a0 = np.linspace(0., 30., Z)
    b0 = np.linspace(0., 1., Z)
    shared = list(shared)
    ref = st.copy()
    rc, iters = orc.OracleModel(md).march(ref, w, a0, b0)
    assert rc == 0
    sm = ShardedMarch(md, 0, 1, device_index=0, collective="native", force_shared=shared, fuse_always=True)
    assert sm.n_shared_zones == len(shared) and sm.batch.n_fused_surfaces > 0
    got = st.copy()
    sm.batch.upload_state(got)
    sm.march_resident(w[:5], a0, b0)
    sm.march_resident(w[5:], a0, b0)
    sm.synchronize()
    assert sm.batch.nomass_iterations() == iters
    assert sm.batch.n_fused_launches > 0
    sm.batch.download_state(got)
    # the torch collective drives the split-phase calls: everything streamed there
    sm2 = ShardedMarch(md, 0, 1, device_index=0, collective="torch", force_shared=shared)
    got2 = st.copy()
    sm2.batch.upload_state(got2)
    sm2.march_resident(w, a0, b0)
    sm2.synchronize()
    sm2.batch.download_state(got2)
    sm.close()
    sm2.close()
    for g in (got, got2):
        for idx in (mdl.node_slots(md), md["hs_front_slot"], md["hs_back_slot"], md["flow_front_slot"],
                    md["flow_back_slot"], md["zone_slot"]):
            assert np.allclose(g[idx], ref[idx], rtol=1e-9, atol=1e-9)


def two_shards_on_one_gpu():
    """The compact zone exchange with two real shards: two batches on one GPU play rank 0 and rank 1, the
    'all-gather' is a concatenation of their partial buffers. Exercises k_zones mode 2 and k_zone_update_shared
    with n_blocks = 2 on the device."""
    from heat_amd import HeatBatch
    from heat_amd.sharded import shard_model, shared_zones
    md, st = mdl.ragged_mixed(900, Z=9, dt=45.0, seed=33)
    w = mdl.weather_series(7, 45.0)
    a0 = np.linspace(0., 50., 9)
    b0 = np.linspace(0., 3., 9)
    ref = st.copy()
    rc, _ = orc.OracleModel(md).march(ref, w, a0, b0)
    assert rc == 0
    stream = torch.cuda.Stream()
    shards = [shard_model(md, r, 2) for r in range(2)]
    batches = [HeatBatch(sh, stream=stream.cuda_stream, n_ranks=2, rank=r) for r, sh in enumerate(shards)]
    masks = [b.touched_zones() for b in batches]
    shared = shared_zones(masks)
    assert 0 < len(shared) < 9
    ns = len(shared)
    parts = [torch.zeros(2 * ns, dtype=torch.float64, device="cuda") for _ in range(2)]
    states = [st.copy(), st.copy()]
    with torch.cuda.stream(stream):
        for b, p, s_ in zip(batches, parts, states):
            b.set_shared_zones(shared)
            b.use_partials(p.data_ptr())
            b.upload_state(s_)
            b.set_weather(w, a0, b0)
        for i in range(len(w)):
            for b in batches:
                b.step_surfaces(i)
            g = torch.cat(parts)
            for b in batches:
                b.step_zones(g.data_ptr(), 2)
        for b, s_ in zip(batches, states):
            b.synchronize()
            b.download_state(s_)
    for r in range(2):
        sh = shards[r]
        for idx in (mdl.node_slots(sh), sh["hs_front_slot"], sh["hs_back_slot"], sh["flow_front_slot"], sh["flow_back_slot"]):
            assert np.allclose(states[r][idx], ref[idx], rtol=1e-9, atol=1e-9)
        mine = np.nonzero(masks[r])[0]
        assert np.allclose(states[r][md["zone_slot"][mine]], ref[md["zone_slot"][mine]], rtol=1e-9, atol=1e-9)
    assert np.array_equal(states[0][md["zone_slot"][shared]], states[1][md["zone_slot"][shared]])
    for b in batches:
        b.close()


if __name__ == "__main__":
    main()
