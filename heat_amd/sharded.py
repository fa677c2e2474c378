"""Multi-GPU plumbing: one process per GPU, surfaces sharded, zones replicated.

Surfaces are independent inside a sub-timestep (reference src/model.rs:102-180); the only
exchange is the zone heat balance ``a[z] = sum h A T_surf``, ``b[z] = sum h A`` over all surfaces
touching zone z (src/model.rs:556-590). Each rank computes the partial sums of its own surfaces.
Zones that only one rank touches are finished right there; for the zones shared between ranks
(agreed on once at setup) the partial blocks are all-gathered over RCCL/xGMI and every rank applies
the zone update from the blocks summed in rank order — so all replicas of a shared zone's
temperature stay bitwise identical and the result does not depend on the collective's internal
reduction order. A rank does not keep zones it never touches up to date.

Two ways to run the collective (``ShardedMarch(collective=...)``):

``"native"`` (default on GPUs)
    the library owns an RCCL communicator (``heat_batch_comm_init``; torch.distributed only carries
    the 128-byte unique id to the ranks) and ``heat_batch_march_resident`` runs kernel -> ncclAllGather
    -> kernel in ONE stream: no second queue, no cross-queue events, no Python per sub-timestep.
    (Measured on MI355X: every dependency between two HIP streams costs 10-28 us even when it is
    already satisfied — as much as the collective itself; see profiles/README.md.)
``"torch"``
    the split-phase C ABI (``heat_batch_step_surfaces`` / ``heat_batch_step_zones``) with
    ``torch.distributed.all_gather_into_tensor`` in between ("nccl" on GPUs, "gloo" on CPU in tests):
    for hosts that bring their own collective.
"""
import numpy as np


def shard_ranges(n_surfaces, n_ranks):
    """Contiguous, balanced surface ranges: rank r owns [bounds[r], bounds[r+1])."""
    base, rem = divmod(int(n_surfaces), int(n_ranks))
    sizes = [base + (1 if r < rem else 0) for r in range(n_ranks)]
    return np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)


def shard_model(md, rank, n_ranks):
    """Model dict of this rank's surfaces; zones, state slots and n_state stay global."""
    from . import modeldict as mdl
    b = shard_ranges(md["n_surfaces"], n_ranks)
    return mdl.subset(md, np.arange(b[rank], b[rank + 1]))


def partition_model(md, n_ranks):
    """heat_partition (host-only, include/heat_amd.h): rank of every surface, whole zone-connected clusters kept
    together — and the number of zones that still end up shared (0: the sharded march needs no collective)."""
    from . import binding
    return binding.partition(md, n_ranks)


def shard_by_ranks(md, ranks, rank):
    """Model dict of the surfaces ``ranks == rank``; zones, state slots and n_state stay global."""
    from . import modeldict as mdl
    return mdl.subset(md, np.nonzero(np.asarray(ranks) == rank)[0])


def touched_mask(md_shard):
    """What heat_batch_touched_zones reports for a shard: 1 for every zone one of its surfaces faces."""
    from . import modeldict as mdl
    t = np.zeros(int(md_shard["n_zones"]), dtype=np.uint8)
    t[np.asarray(md_shard["front_zone"])[np.asarray(md_shard["front_kind"]) == mdl.SPACE]] = 1
    t[np.asarray(md_shard["back_zone"])[np.asarray(md_shard["back_kind"]) == mdl.SPACE]] = 1
    return t


def zone_roles(touch_count, local_mask, rank, n_ranks):
    """From the all-reduced touch counts: (shared zones — faced from two ranks or more, the only ones exchanged;
    owned mask — zones this rank finishes: those it faces plus, of the zones NO rank faces, every n_ranks-th one:
    such a zone still follows its a0 / b0 terms, reference src/model.rs:410-423)."""
    cnt = np.asarray(touch_count)
    z = np.arange(len(cnt))
    shared = np.nonzero(cnt >= 2)[0].astype(np.int32)
    owned = (np.asarray(local_mask) != 0) | ((cnt == 0) & (z % n_ranks == rank))
    return shared, owned.astype(np.uint8)


def shared_zones(touched_masks):
    """touched_masks: [n_ranks, n_zones] 0/1. Returns the sorted global numbers of the zones that more than one
    rank touches (the only zones whose heat balance needs an exchange)."""
    m = np.asarray(touched_masks)
    return np.nonzero(m.astype(np.int64).sum(axis=0) >= 2)[0].astype(np.int32)


class ZoneExchange:
    """All-gather of the per-rank partial (a, b) blocks of the SHARED zones:
    [2 * n_shared] per rank -> [n_ranks, 2 * n_shared] on every rank (block order = rank order)."""

    def __init__(self, n_shared, device, group=None):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_shared = int(n_shared)
        n = max(2 * self.n_shared, 2)
        self.partial = torch.zeros(n, dtype=torch.float64, device=device)
        self.gathered = torch.zeros(self.world * n, dtype=torch.float64, device=device)

    def all_gather(self):
        if self.n_shared == 0:
            return self.gathered
        if self.world == 1:
            self.gathered.copy_(self.partial)
        else:
            self.dist.all_gather_into_tensor(self.gathered, self.partial, group=self.group)
        return self.gathered


def agree_on_zones(local_mask, device, group=None):
    """Every rank contributes its touched-zone mask; all get the same shared-zone list, and each its owned mask
    (zone_roles)."""
    import torch
    import torch.distributed as dist
    m = torch.as_tensor(np.asarray(local_mask, dtype=np.int32), device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)
        return zone_roles(m.cpu().numpy(), local_mask, dist.get_rank(group), dist.get_world_size(group))
    return np.zeros(0, dtype=np.int32), np.ones(len(local_mask), dtype=np.uint8)


def agree_on_shared_zones(local_mask, device, group=None):
    """The shared-zone list alone (agree_on_zones)."""
    return agree_on_zones(local_mask, device, group)[0]


def native_comm_or_fallback(comm_init, comm_destroy, agree_min, log=None):
    """The collective decision around heat_batch_comm_init: ``comm_init()`` is tried on every rank (it raises on
    failure — HEAT_E_COMM: ncclCommInitRank refused, a rank without RCCL, ...); the ranks then agree on the WORST
    outcome (``agree_min(ok) -> min over ranks``, an all-reduce over the host's own process group). When any rank
    failed, the ranks that did get a communicator give it up (``comm_destroy()``), so that EVERY rank takes the same
    fallback. Returns (True, None) when all ranks hold a communicator, else (False, message of this rank's failure or
    a note that another rank failed). Never raises for a failed comm_init: a sharded run must not lose a rank there."""
    err = None
    try:
        comm_init()
        ok = 1
    except Exception as e:  # noqa: BLE001 — HeatError, or whatever a host's own comm_init raises
        ok = 0
        err = "%s: %s" % (type(e).__name__, e)
    all_ok = int(agree_min(ok))
    if all_ok:
        return True, None
    if ok:
        comm_destroy()
        err = "heat_batch_comm_init failed on another rank"
    if log:
        log("heat_amd: native RCCL communicator not available on every rank (%s); falling back" % err)
    return False, err


class ShardedMarch:
    """Drives one rank's HeatBatch through the sharded sub-timestep with the zone exchange.

    ``md`` is this rank's shard — or, with ``rank_of_surface`` (partition_model), the WHOLE model, from which the
    library picks the rank's surfaces (heat_batch_create_shard). A partition that shares no zone (``n_shared == 0``
    from partition_model, the usual case: clusters are kept whole) needs no communicator and no collective at all:
    every rank marches its clusters on its own (``collective`` ends up "none")."""

    def __init__(self, md, rank, n_ranks, device_index=0, collective="native", force_shared=None,
                 rank_of_surface=None, n_shared_in_partition=None, **batch_opts):
        import sys
        from . import binding
        if binding._lib is not None and "torch" not in sys.modules:
            raise RuntimeError(
                "heat_amd: libheat_amd.so was loaded before torch; torch ships its own HIP runtime and the two "
                "cannot share a device in one process. Import torch before creating the first HeatBatch.")
        import torch
        import torch.distributed as dist
        from .binding import HeatBatch, comm_available, comm_unique_id
        if collective not in ("native", "torch"):
            raise ValueError("collective must be 'native' or 'torch'")
        self.torch = torch
        self.collective = collective
        self.n_ranks = n_ranks
        torch.cuda.set_device(device_index)
        dev = torch.device("cuda", device_index)
        # A dedicated non-default stream: the library runs its kernels (and, natively, the collective) on it;
        # torch orders its own collective against it (the legacy default stream has handle 0, which the C ABI
        # reads as "create your own stream").
        self.stream = torch.cuda.Stream(device=device_index)
        self.batch = HeatBatch(md, device=device_index, stream=self.stream.cuda_stream, n_ranks=n_ranks,
                               rank=rank, rank_of_surface=rank_of_surface, **batch_opts)
        self.shared = None
        self.exchange = None
        self.collective_fallback = None  # why "native" was asked for and something else runs (None: it does not apply)
        forced = None if force_shared is None or len(force_shared) == 0 else np.asarray(force_shared, dtype=np.int32)
        if rank_of_surface is not None and n_shared_in_partition == 0 and forced is None:
            # heat_batch_create_shard has seen that no zone is faced from two ranks: nothing to exchange, ever
            self.collective = "none"
            return
        multi = dist.is_initialized() and dist.get_world_size() > 1
        on_dev = multi and dist.get_backend() == "nccl"
        if collective == "native":
            # The ranks vote on RCCL BEFORE any of them enters the collective ncclCommInitRank: a rank that cannot
            # load it would leave the others waiting inside.
            ok = 1 if comm_available() else 0
            if multi:
                t = torch.tensor([ok], dtype=torch.int32, device=dev if on_dev else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                ok = int(t.item())
            if ok:
                # rank 0 draws the RCCL unique id; torch.distributed only carries its 128 bytes to the others
                uid = torch.zeros(128, dtype=torch.uint8)
                if rank == 0:
                    uid = torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8).clone()
                if multi:
                    t = uid.to(dev) if on_dev else uid
                    dist.broadcast(t, src=0)
                    uid = t.cpu()
                # the zones the ranks share are agreed inside (plus the forced ones: the union, on every rank)
                def agree_min(v):
                    if not multi:
                        return v
                    t = torch.tensor([v], dtype=torch.int32, device=dev if on_dev else "cpu")
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    return int(t.item())

                done, why = native_comm_or_fallback(
                    lambda: self.batch.comm_init(uid.numpy().tobytes(), extra_shared=forced),
                    self.batch.comm_destroy, agree_min, log=lambda m: print(m, file=sys.stderr))
                if done:
                    return
                self.collective_fallback = why
            else:
                self.collective_fallback = "RCCL cannot be loaded on every rank"
                print("heat_amd: RCCL cannot be loaded on every rank; using torch.distributed for the zone exchange",
                      file=sys.stderr)
            self.collective = "torch"
        self.shared, owned = agree_on_zones(self.batch.touched_zones(), dev)
        if forced is not None:
            # tests / single-GPU rehearsal: treat these zones as shared although no other rank touches them
            self.shared = np.union1d(self.shared, forced).astype(np.int32)
        self.batch.set_owned_zones(owned)
        self.batch.set_shared_zones(self.shared)
        self.exchange = ZoneExchange(len(self.shared), dev)
        self.batch.use_partials(self.exchange.partial.data_ptr())

    @property
    def n_shared_zones(self):
        return self.batch.n_shared_zones

    @property
    def comm_ranks(self):
        """Ranks of the library-owned RCCL communicator (0: none — no zone shared, or the torch collective)."""
        return self.batch.comm_ranks

    def march_resident(self, weather, zone_a0=None, zone_b0=None):
        """≙ ThermalModel::march on the device-resident state of this shard (asynchronous)."""
        b = self.batch
        if self.collective in ("native", "none"):
            b.march_resident(weather, zone_a0, zone_b0)
            return
        with self.torch.cuda.stream(self.stream):
            b.set_weather(weather, zone_a0, zone_b0)
            for i in range(len(weather)):
                b.step_surfaces(i)      # local kernels; zones only this rank touches are updated in place
                if len(self.shared):
                    g = self.exchange.all_gather()
                    b.step_zones(g.data_ptr(), self.exchange.world)

    def synchronize(self):
        self.batch.synchronize()

    def close(self):
        self.batch.close()
