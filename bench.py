#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X wall heat-conduction path.

A *step* is one sub-timestep of ThermalModel::march (reference src/model.rs:369-424) over the
whole batch: iterate_surfaces for every surface + the zone update. The workload is the
north_star headline of BASELINE.json: 1 000 000 all-massive surfaces x 32 nodes per GPU
(weak scaling: every rank holds its own million), RK4 + convection / long-wave / solar boundary
updates, zones of 100 surfaces. State is resident in HBM when the timed region starts.

The K timed steps are issued as march calls of --substeps-per-march sub-timesteps each (default 20:
a 15-minute model timestep at dt = 45 s), as ThermalModel::march runs its dt_subdivisions
sub-timesteps per call. Inside one call the library keeps zone-connected clusters of surfaces
resident on the chip (cluster-resident march, include/heat_amd.h); --no-fusion streams every
sub-timestep through HBM instead, and a second, shorter leg always measures that streamed kernel
(`roofline_streaming`) so that the per-sub-timestep HBM roofline stays on record.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      the surface kernel's algorithmic GB/s (HIP events on the kernel's stream, recorded
                inside the timed region) against the 8 TB/s HBM3E peak;
  cpu_baseline  the CPU oracle (oracle/, a C port of the reference path) timed on this box's host
                cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def build_shard(S, n, zones_per_gpu, rank, world, dt, seed):
    """This rank's million walls. Zones are global; zone boundaries are offset by half a zone so
    that the first/last zone of every rank also has surfaces on the neighbouring rank."""
    from heat_amd import modeldict as mdl
    Z = zones_per_gpu * world
    md, state = mdl.uniform_massive(S, n, Z=Z, dt=dt, seed=seed + rank)
    g = rank * S + np.arange(S, dtype=np.int64)
    per_zone = max(1, (S * world) // Z)
    md["back_zone"] = (((g + per_zone // 2) // per_zone) % Z).astype(np.int32)
    md["front_zone"] = md["back_zone"].copy()
    return md, state


CONFIGS = ("headline", "2", "3", "5", "partitions")


def build_config(name, args, rank, world, dt, seed):
    """The workloads of BASELINE.json / SURVEY.md §8(d). Returns (model dict, initial state, description)."""
    from heat_amd import modeldict as mdl
    if name == "headline":
        md, state = build_shard(args.surfaces, args.nodes, args.zones_per_gpu, rank, world, dt, seed)
        return md, state, ("north_star headline: %d all-massive surfaces x %d nodes per GPU, RK4 + TARP convection + "
                           "long-wave + solar boundaries, %d zones per GPU, dt = %g s" % (
                               args.surfaces, args.nodes, args.zones_per_gpu, dt))
    if name == "2":
        md, state = mdl.uniform_massive(10_000, 20, Z=100, dt=90.0, identical=True, vertical=True)
        return md, state, "BASELINE config 2: 10 000 identical 3-layer massive walls x 20 nodes, 100 zones, dt = 90 s"
    if name == "3":
        S = args.surfaces
        md, state = mdl.ragged_mixed(S, Z=max(1, S // 100), dt=dt, seed=seed)
        return md, state, ("BASELINE config 3: %d ragged surfaces of 8-64 nodes (70 %% massive, 20 %% massive core "
                           "between no-mass facings, 10 %% two-node no-mass), mixed boundaries, %d zones joined in a "
                           "ring by the Space/Space walls, dt = %g s" % (S, max(1, S // 100), dt))
    if name == "5":
        S = 200_000 if args.surfaces == 1_000_000 else args.surfaces
        md, state = mdl.glazing_cavity(S, Z=max(1, S // 100), dt=dt)
        return md, state, ("BASELINE config 5: %d surfaces, half double glazing (4 no-mass nodes around a gas cavity), "
                           "half Trombe-like (concrete / air cavity / glass, 17 nodes), dt = %g s" % (S, dt))
    raise SystemExit("unknown --config %r" % name)


def cpu_baseline(n, dt, seed, target_seconds=12.0):
    """Times the oracle (single thread: the reference is single-threaded, model.rs:113-116) on a
    bounded sample of the same workload."""
    from heat_amd import modeldict as mdl
    from oracle import oracle as orc
    S_cpu, steps = 20000, 10
    md, state = mdl.uniform_massive(S_cpu, n, Z=S_cpu // 100, dt=dt, seed=seed)
    m = orc.OracleModel(md)
    w = mdl.weather_series(steps, dt)
    t0 = time.perf_counter()
    rc, _ = m.march(state, w)
    t1 = time.perf_counter() - t0
    assert rc == 0
    rate = S_cpu * n * steps / t1
    # scale the sample up to ~target_seconds of CPU work
    steps2 = int(max(steps, min(400, target_seconds * rate / (S_cpu * n))))
    w = mdl.weather_series(steps2, dt)
    t0 = time.perf_counter()
    rc, _ = m.march(state, w)
    t2 = time.perf_counter() - t0
    assert rc == 0
    out = {"value": S_cpu * n * steps2 / t2, "unit": "node-updates/s", "cores": 1, "kind": "port",
           "sample": "%d surfaces x %d nodes x %d sub-timesteps of the same workload, %.1f s; "
                     "tri-diagonal storage, no per-step allocation: an upper bound on the Rust reference's speed"
                     % (S_cpu, n, steps2, t2)}
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = min(ncores, 16)  # the GPU box gives one GPU's share of the host: 16 cores
    if ncores > 1:
        st2 = state.copy()
        t0 = time.perf_counter()
        rc, _ = m.march(st2, w[:max(1, steps2 // 2)], threads=ncores)
        t3 = time.perf_counter() - t0
        out["all_cores"] = {"value": S_cpu * n * max(1, steps2 // 2) / t3, "cores": ncores,
                            "note": "OpenMP over surfaces = the reference's disabled rayon path (model.rs:113-116)"}
    return out


def pmc_traffic(surfaces, nodes, mode="streamed", substeps=1):
    """HBM bytes per launch of the surface kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE; bench.py cannot run the profiler on itself). None when no
    committed measurement matches this workload and kernel (mode: "streamed" = one sub-timestep per launch,
    "fused" = `substeps` sub-timesteps per launch)."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith("_pmc_traffic.json"):
            try:
                j = json.load(open(os.path.join(pdir, f)))
            except Exception:
                continue
            wl = j.get("workload", {})
            if wl.get("surfaces") != surfaces or wl.get("nodes") != nodes or wl.get("mode", "streamed") != mode:
                continue
            if mode == "fused" and wl.get("substeps_per_launch") != substeps:
                continue
            best = (j["traffic_bytes_per_launch"], "profiles/" + f)
    return best if best else (None, None)


def march_in_calls(march, weather, per_call):
    """K sub-timesteps as march calls of `per_call` sub-timesteps (ThermalModel::march = one call)."""
    for i in range(0, len(weather), per_call):
        march(weather[i:i + per_call])


def streaming_leg(md, state, args, dt, steps=60, warmup=10):
    """The per-sub-timestep kernel on its own: a batch planned without the cluster-resident march
    (16 nodes per lane, persistent waves), HIP events around every launch."""
    from heat_amd import HeatBatch, modeldict as mdl
    with HeatBatch(md, nodes_per_lane=args.nodes_per_lane, no_palette=args.no_palette, no_fusion=True) as b:
        b.upload_state(state)
        b.march_resident(mdl.weather_series(warmup, dt))
        b.synchronize()
        b.set_timing(True)
        t0 = time.perf_counter()
        b.march_resident(mdl.weather_series(steps, dt, t0=dt * warmup))
        b.synchronize()
        el = time.perf_counter() - t0
        surf_us, substep_us, n = b.get_timing()
        ab = b.algorithmic_bytes
        counts = b.class_counts()
    achieved = ab / (surf_us * 1e-6) / 1e9
    traffic, src = pmc_traffic(args.surfaces, args.nodes, "streamed")
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src,
            "kernel": "k_surfaces_fast<M,...,0> (iterate_surfaces, one sub-timestep per launch, state streamed through HBM)",
            "algorithmic_bytes_per_launch": ab, "kernel_us": surf_us, "substep_us": substep_us, "samples": n,
            "node_updates_per_sec": int(md["node_offset"][-1]) * steps / el,
            "kernel_classes[M4,M8,M16,small,general]": counts}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=CONFIGS, default="headline",
                    help="workload: the north_star headline (default) or a BASELINE.json config")
    ap.add_argument("--surfaces", type=int, default=1_000_000, help="surfaces per GPU")
    ap.add_argument("--nodes", type=int, default=32)
    ap.add_argument("--zones-per-gpu", type=int, default=10_000)
    ap.add_argument("--nodes-per-lane", type=int, default=0)
    ap.add_argument("--no-palette", action="store_true", help="keep dt/mass and U as per-node arrays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--substeps-per-march", type=int, default=20,
                    help="sub-timesteps per march call (ThermalModel::march runs dt_subdivisions of them per call)")
    ap.add_argument("--no-fusion", action="store_true",
                    help="stream every sub-timestep through HBM (no cluster-resident march)")
    ap.add_argument("--no-streaming-leg", action="store_true", help="skip the second leg that measures the streamed kernel")
    ap.add_argument("--no-timing", action="store_true", help="do not record HIP events in the timed region")
    ap.add_argument("--force-sharded", action="store_true",
                    help="drive the multi-GPU code path (ShardedMarch + zone exchange) even with one rank")
    ap.add_argument("--force-shared-zones", type=int, default=0,
                    help="single-GPU rehearsal of the exchange: declare this many zones shared (implies --force-sharded)")
    ap.add_argument("--collective", choices=("native", "torch"), default="native",
                    help="sharded: library-owned RCCL communicator, all in one stream (default), or "
                         "torch.distributed.all_gather_into_tensor between the split-phase calls")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON result: anything a library prints meanwhile (RCCL's version
    # banner under NCCL_DEBUG=VERSION goes to stdout) is sent to stderr instead.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    K, W = args.steps, args.warmup
    dt = 45.0
    seed = 20260401

    from heat_amd import HeatBatch, modeldict as mdl
    md, state, workload = build_config(args.config, args, rank, world, dt, seed)
    dt = float(md["dt"])
    n_nodes_local = int(md["node_offset"][-1])
    weather_w = mdl.weather_series(max(W, 1), dt)
    weather_k = mdl.weather_series(K, dt, t0=dt * W)

    sharded = world > 1 or args.force_sharded or args.force_shared_zones > 0
    if sharded:
        import torch
        import torch.distributed as dist
        from heat_amd.sharded import ShardedMarch
        torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        forced = None
        if args.force_shared_zones > 0:
            forced = np.unique(np.linspace(0, args.zones_per_gpu * world - 1, args.force_shared_zones).astype(np.int32))
        sm = ShardedMarch(md, rank, world, device_index=local_rank, collective=args.collective, force_shared=forced,
                          nodes_per_lane=args.nodes_per_lane, no_palette=args.no_palette, no_fusion=args.no_fusion)
        batch = sm.batch
        batch.upload_state(state)

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()
            batch.synchronize()  # reports device-side numerical flags

        run = sm.march_resident
    else:
        batch = HeatBatch(md, device=local_rank, nodes_per_lane=args.nodes_per_lane, use_graph=True,
                          no_palette=args.no_palette, no_fusion=args.no_fusion)
        batch.upload_state(state)

        def barrier():
            batch.synchronize()

        run = batch.march_resident

    P = max(1, args.substeps_per_march)
    if W > 0:
        march_in_calls(run, weather_w[:W], P)
    barrier()
    batch.set_timing(not args.no_timing)
    t0 = time.perf_counter()
    march_in_calls(run, weather_k, P)
    barrier()
    elapsed = time.perf_counter() - t0
    surf_us, substep_us, n_samples = batch.get_timing()
    batch.set_timing(False)

    if sharded:
        import torch
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    algorithmic_bytes = batch.algorithmic_bytes
    counts = batch.class_counts()
    n_fused = batch.n_fused_surfaces if not args.no_fusion else 0
    fused = n_fused > 0
    total_nodes = n_nodes_local * world
    value = total_nodes * K / elapsed
    result = {
        "metric": "surface-node-updates/sec",
        "value": value,
        "unit": "node-updates/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "sub_timesteps_per_sec": K / elapsed,
        "config": {
            "workload": workload + "; one step = one sub-timestep (iterate_surfaces + zone update)",
            "config": args.config,
            "surfaces_per_gpu": int(md["n_surfaces"]), "nodes_per_gpu": n_nodes_local,
            "zones_per_gpu": int(md["n_zones"]), "dt_s": dt, "substeps_per_march": P,
            "kernel_classes[M4,M8,M16,small,general]": counts,
            "surfaces_in_cluster_resident_march": n_fused,
            "parallelism": "surfaces sharded %d-way, zones replicated, per-sub-timestep RCCL all-gather of the partial "
                           "sums of %d shared zones (%s)" % (
                               world, sm.n_shared_zones,
                               "library-owned communicator, one stream" if args.collective == "native" else "torch.distributed")
            if sharded else "single GPU",
        },
    }
    if n_samples > 0:
        # algorithmic bytes of a launch = SURVEY.md §8(d)'s 32 n + 152 bytes per surface and sub-timestep x the
        # sub-timesteps the launch marches (1 streamed; P cluster-resident)
        per_launch = P if fused else 1
        achieved = algorithmic_bytes / (surf_us * 1e-6) / 1e9
        traffic, traffic_src = pmc_traffic(args.surfaces, args.nodes, "fused" if fused else "streamed", per_launch)
        result["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": ("k_surfaces_fast<M,...,FUSED> (cluster-resident march: %d sub-timesteps of iterate_surfaces + zone "
                       "update per launch, node temperatures in registers; the streamed kernel's own line is "
                       "roofline_streaming)" % P) if fused else
                      "k_surfaces_fast (iterate_surfaces: RK4 stencil + boundary updates, one sub-timestep per launch)",
            "sub_timesteps_per_launch": per_launch,
            "algorithmic_bytes_per_launch": algorithmic_bytes * per_launch,
            "kernel_us": surf_us * per_launch, "kernel_us_per_sub_timestep": surf_us, "substep_us": substep_us,
            "samples": n_samples,
            "note": ("frac > 1 is possible: the launch re-uses the state on chip instead of streaming it per "
                     "sub-timestep, so its HBM traffic is far below the algorithmic (streaming) byte count") if fused else None,
        }
    if rank == 0 and world == 1 and fused and not args.no_streaming_leg:
        batch.close()
        result["roofline_streaming"] = streaming_leg(md, state, args, dt)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.nodes, dt, seed)
    if sharded:
        import torch.distributed as dist
        sm.close()
        dist.destroy_process_group()
    else:
        batch.close()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
