#!/usr/bin/env python
"""bench.py — benchmark of the MI355X wall heat-conduction path.

A *step* is one sub-timestep of ThermalModel::march (reference src/model.rs:369-424) over the whole
model: iterate_surfaces for every surface + the zone update. The default workload is the north_star
headline of BASELINE.json — 1 000 000 all-massive surfaces x 32 nodes, RK4 + convection / long-wave /
solar boundary updates, zones of 100 surfaces. State is resident in HBM when the timed region starts.

The K timed steps are issued as march calls of --substeps-per-march sub-timesteps each (default 20: a
15-minute model timestep at dt = 45 s), as ThermalModel::march runs its dt_subdivisions sub-timesteps
per call. Inside one call the library keeps zone-connected clusters of surfaces resident on the chip
(cluster-resident march, include/heat_amd.h) where its planner expects a gain; --no-fusion streams
every sub-timestep through HBM instead, and a second, shorter leg always measures that streamed kernel
(`roofline_streaming`) so that the per-sub-timestep HBM roofline stays on record.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

With --config headline (the default) and one GPU the run goes on, after the headline, through the other BASELINE.json
configs as short legs of their own — config 3 (1 M ragged mixed walls), config 5 (glazing + cavities; `5x10`: 1 M + 1 M
of them), config 2 (10 000 identical walls), `partitions` (buildings of 8 rooms joined by interior walls) and
`buildings40` (of 40 rooms: clusters larger than a workgroup, marched by teams of workgroups) — and attaches each as
`"configs": {"3": {value, ms_per_step, roofline{...}, cpu_baseline{...}}, ...}` to the SAME JSON line
(--no-configs skips them; `--config 3` etc. runs one of them as the main leg).

N > 1 (BASELINE config 4) shards the SAME model over the ranks ("scaling": "strong"): heat_partition cuts it
along its zone-connected clusters, so that no zone is shared and no collective is issued (the headline). A second,
short leg then runs config 3, whose ring of zones is ONE cluster: it is cut by surface ranges and its cut zones are
exchanged with an RCCL all-gather per sub-timestep on the library's own communicator — `"exchange_leg"` reports
n_shared_zones, collective, comm_ranks (and collective_fallback when the native communicator could not be had on
every rank). `--scaling weak` gives every rank a model of its own with zones shared across the rank boundaries.

Before the W warmup steps of every leg the batch marches untimed for --settle-ms (60 ms; `settle` in the line): the chip needs
20-30 ms of load to reach the clock it then holds (tools/settle.py: the headline's sub-timestep goes 69.8 -> 64.4 us over the
first 25 ms, and is back at 71 after two idle seconds) — a timed region of one march call would otherwise measure that ramp.

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline            the dominant kernel against the resource that bounds it. Streamed march: "hbm" — algorithmic
                      bytes / kernel time (HIP events on the kernel's stream) against 8 TB/s. Cluster-resident march:
                      "valu_f64" — ALGORITHMIC f64 flops (40 per node-update) / kernel time against
                      the f64 vector peak, 78.6 TFLOP/s; its `issue` sub-object keeps the issue-slot view (measured
                      instruction counts by class from the committed PMC passes x cycles per class) and is nulled,
                      with "counters_stale": true, when the committed counters were taken on other kernel sources;
                      Streamed legs carry `kernel_us_rocprof` too (the kernels' mean durations in the committed trace of the
                      same sources: HIP events around a two-kernel sub-timestep include the gap between its launches) and
                      the `issue` view; BASELINE config 5, which moves a tenth of the bytes the chip could, names the
                      instruction issue of its dependent chains as its bound ("valu_issue") and keeps the HBM figures beside;
  roofline_streaming  the streamed kernel's own line whenever the main run was cluster-resident;
  cpu_baseline        the CPU oracle (oracle/, a C port of the reference path) timed on this box's host cores on a
                      bounded sample of the same workload (rank 0, N = 1 only);
  caller_owned        heat_batch_march on a caller-owned host state (the drop-in call of ThermalModel::march):
                      PCIe-inclusive rate, never `value`.
"""
import argparse
import hashlib
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# f64 vector peak: 256 CUs x 4 SIMDs x 16 f64 lanes per cycle x 2 flops (FMA) x 2.4 GHz (a wave64 f64 instruction
# issues over 4 cycles; MI355X_MICROARCH.md "Wave scheduling", 32-bit VALU: 2 cycles)
F64_PEAK_TFLOPS = 256 * 4 * 16 * 2 * 2.4e9 / 1e12
SIMD_CYCLES_PER_SEC = 256 * 4 * 2.4e9
VALU_CYCLES_F64, VALU_CYCLES_OTHER = 4.0, 2.0
# ~ 4 (5 + 2) + 12 = 40 flops per node-update of the RK4 stencil (boundary work is O(1) per surface)
ALGORITHMIC_FLOPS_PER_NODE_UPDATE = 40.0

CONFIGS = ("headline", "2", "3", "5", "partitions", "buildings40", "5x10")
KERNEL_SOURCES = ("heat_amd/csrc/kernels.hip", "heat_amd/csrc/device_math.hpp", "heat_amd/csrc/layout.hpp")


def kernel_source_hash():
    """sha256 over the device sources: committed counters are only good for the kernels they were measured on
    (tools/make_counters.py stamps the same hash into profiles/*_counters.json)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build_weak_shard(S, n, zones_per_gpu, rank, world, dt, seed):
    """--scaling weak: this rank's own walls. Zones are global; zone boundaries are offset by half a zone so that
    the first/last zone of every rank also has surfaces on the neighbouring rank (the exchange really runs)."""
    from heat_amd import modeldict as mdl
    Z = zones_per_gpu * world
    md, state = mdl.uniform_massive(S, n, Z=Z, dt=dt, seed=seed + rank)
    g = rank * S + np.arange(S, dtype=np.int64)
    per_zone = max(1, (S * world) // Z)
    md["back_zone"] = (((g + per_zone // 2) // per_zone) % Z).astype(np.int32)
    md["front_zone"] = md["back_zone"].copy()
    return md, state


def build_config(name, args, dt, seed):
    """The workloads of BASELINE.json / SURVEY.md §8(d), whole model. Returns (model dict, state, description)."""
    from heat_amd import modeldict as mdl
    S = args.surfaces
    if name == "headline":
        Z = max(1, S // 100)
        md, state = mdl.uniform_massive(S, args.nodes, Z=Z, dt=dt, seed=seed)
        return md, state, ("north_star headline: %d all-massive surfaces x %d nodes, RK4 + TARP convection + long-wave + "
                           "solar boundaries, %d zones of 100 walls each (every zone a cluster of its own), dt = %g s"
                           % (S, args.nodes, Z, dt))
    if name == "2":
        md, state = mdl.uniform_massive(10_000, 20, Z=100, dt=90.0, identical=True, vertical=True)
        return md, state, "BASELINE config 2: 10 000 identical 3-layer massive walls x 20 nodes, 100 zones, dt = 90 s"
    if name == "3":
        md, state = mdl.ragged_mixed(S, Z=max(1, S // 100), dt=dt, seed=seed)
        return md, state, ("BASELINE config 3: %d ragged surfaces of 8-64 nodes (70 %% massive, 20 %% massive core "
                           "between no-mass facings, 10 %% two-node no-mass), mixed boundaries, %d zones joined in a "
                           "ring by the Space/Space walls (one cluster), dt = %g s" % (S, max(1, S // 100), dt))
    if name == "5":
        S5 = 200_000 if S == 1_000_000 else S
        md, state = mdl.glazing_cavity(S5, Z=max(1, S5 // 100), dt=dt)
        return md, state, ("BASELINE config 5: %d surfaces, half double glazing (4 no-mass nodes around a gas cavity), "
                           "half Trombe-like (concrete / air cavity / glass, 17 nodes), dt = %g s" % (S5, dt))
    if name == "partitions":
        md, state = mdl.partitioned_buildings(S, args.nodes, dt=dt, seed=seed)
        return md, state, ("buildings with interior partitions: %d all-massive walls x %d nodes in buildings of 8 rooms "
                           "x 12 walls, 2 walls of every room are partitions to neighbouring rooms (front and back both "
                           "Space, different zones): clusters of 8 zones, dt = %g s" % (S, args.nodes, dt))
    if name == "buildings40":
        md, state = mdl.partitioned_buildings(S, args.nodes, rooms=40, dt=dt, seed=seed)
        return md, state, ("buildings of 40 rooms: %d all-massive walls x %d nodes, 12 walls per room, 2 of them partitions to "
                           "neighbouring rooms: clusters of 40 zones and 480 walls — larger than a workgroup, marched by teams "
                           "of workgroups (layout.hpp, FusedSuper), dt = %g s" % (int(md["n_surfaces"]), args.nodes, dt))
    if name == "5x10":
        S5 = 2 * S
        md, state = mdl.glazing_cavity(S5, Z=max(1, S5 // 100), dt=dt)
        return md, state, ("BASELINE config 5 at ten times its size: %d double glazings + %d Trombe-like walls, dt = %g s"
                           % (S5 // 2, S5 // 2, dt))
    raise SystemExit("unknown --config %r" % name)


def cpu_baseline(md_full, state_full, dt, target_seconds=12.0):
    """Times the oracle (single thread: the reference is single-threaded, model.rs:113-116) on a bounded sample
    of the same workload: the first surfaces of the model with their zones."""
    from heat_amd import modeldict as mdl
    from oracle import oracle as orc
    S_all = int(md_full["n_surfaces"])
    S_cpu = min(S_all, 20000)
    md = mdl.subset(md_full, np.arange(S_cpu))
    state = state_full.copy()
    nodes = int(md["node_offset"][-1])
    steps = 10 if target_seconds >= 8 else 4
    m = orc.OracleModel(md)
    w = mdl.weather_series(steps, dt)
    t0 = time.perf_counter()
    rc, _ = m.march(state, w)
    t1 = time.perf_counter() - t0
    assert rc == 0
    rate = nodes * steps / t1
    steps2 = int(max(steps, min(400, target_seconds * rate / nodes)))
    w = mdl.weather_series(steps2, dt)
    t0 = time.perf_counter()
    rc, _ = m.march(state, w)
    t2 = time.perf_counter() - t0
    assert rc == 0
    out = {"value": nodes * steps2 / t2, "unit": "node-updates/s", "cores": 1, "kind": "port",
           "sample": "the first %d surfaces (%d nodes) of the same workload x %d sub-timesteps, %.1f s; "
                     "tri-diagonal storage, no per-step allocation: an upper bound on the Rust reference's speed"
                     % (S_cpu, nodes, steps2, t2)}
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = min(ncores, 16)  # the GPU box gives one GPU's share of the host: 16 cores
    if ncores > 1:
        st2 = state.copy()
        half = max(1, steps2 // 2)
        t0 = time.perf_counter()
        rc, _ = m.march(st2, w[:half], threads=ncores)
        t3 = time.perf_counter() - t0
        out["all_cores"] = {"value": nodes * half / t3, "cores": ncores,
                            "note": "OpenMP over surfaces = the reference's disabled rayon path (model.rs:113-116)"}
    return out


def committed_counters(config, surfaces, nodes_total, mode, substeps=1):
    """Counters of the surface kernel from the rocprofv3 PMC passes committed under profiles/ (bench.py cannot run
    the profiler on itself): the newest profiles/*_counters.json whose workload matches. Returns (dict, source, stale):
    stale = the file carries no hash of the kernel sources, or another one than the sources this run was built from."""
    best = (None, None, None)
    pdir = os.path.join(ROOT, "profiles")
    now = kernel_source_hash()
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not f.endswith("_counters.json"):
            continue
        try:
            j = json.load(open(os.path.join(pdir, f)))
        except Exception:
            continue
        wl = j.get("workload", {})
        if wl.get("config") != config or wl.get("surfaces") != surfaces or wl.get("mode") != mode:
            continue
        if wl.get("nodes_total") not in (None, nodes_total):
            continue
        if mode == "fused" and wl.get("substeps_per_launch") != substeps:
            continue
        best = (j, "profiles/" + f, j.get("kernel_sources_sha256") != now)
    return best


def rocprof_kernel_us(counters_src, kernel_substr):
    """Mean duration of the streamed surface kernel(s) of a sub-timestep from the rocprofv3 --kernel-trace --stats summary
    committed beside a counters file (tools/profile_bundle.sh writes both in one go: profiles/<tag>_<name>_kernel_stats.csv
    next to profiles/<tag>_<name>_<mode>_counters.json): the sum over the variants whose name holds `kernel_substr`.
    HIP events around a sub-timestep's launches book the few microseconds between them as kernel time; this does not."""
    if not counters_src:
        return None
    import csv
    base = os.path.basename(counters_src)
    parts = base.split("_")          # r03_cfg3_streamed_counters.json -> r03_cfg3_kernel_stats.csv
    f = os.path.join(ROOT, "profiles", "_".join(parts[:-2]) + "_kernel_stats.csv")
    if not os.path.exists(f):
        return None
    tot = 0.0
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["Name"]:
            tot += float(r["AverageNs"]) / 1e3
    return tot or None


SETTLE_NOTE = ("untimed march calls of the same workload before the W warmup steps: the chip takes 20-30 ms of load to reach "
               "the clock it then holds (tools/settle.py: 69.8 -> 64.4 us per sub-timestep of the headline over the first 25 ms; "
               "back to 71 after two idle seconds) — without them a short timed region measures the ramp, not the kernel")


def settle(march, weather, sync, ms, n_calls=None):
    """The settle phase: march `weather` (one call) again and again for `ms` milliseconds, or exactly n_calls times
    (N > 1: every rank the same count, agreed by the caller). Returns (calls, seconds)."""
    t0 = time.perf_counter()
    n = 0
    while (n < n_calls) if n_calls is not None else ((time.perf_counter() - t0) * 1e3 < ms):
        march(weather)
        sync()
        n += 1
    return n, time.perf_counter() - t0


def march_in_calls(march, weather, per_call):
    """K sub-timesteps as march calls of `per_call` sub-timesteps (ThermalModel::march = one call)."""
    for i in range(0, len(weather), per_call):
        march(weather[i:i + per_call])


def issue_view(counters, per_sub, surf_us):
    """Instruction issue from the committed per-class counts of a workload: wave-instructions x cycles per class
    (f64 4, every other VALU instruction 2) / (1024 SIMDs x 2.4 GHz x kernel time); `per_sub` scales the file's
    per-launch counts to one sub-timestep of this run."""
    valu = counters["valu_insts_per_launch"] * per_sub
    issue = {"valu_insts_per_sub_timestep": valu}
    cl = counters.get("counters_per_launch", {})
    f64 = [cl.get(k) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")]
    meas = None
    if all(v is not None for v in f64):
        n64 = sum(f64) * per_sub
        cyc = n64 * VALU_CYCLES_F64 + max(valu - n64, 0.0) * VALU_CYCLES_OTHER
        issue.update({"f64_insts_per_sub_timestep": n64, "issue_cycles_per_sub_timestep": cyc,
                      "frac": cyc / (surf_us * 1e-6) / SIMD_CYCLES_PER_SEC,
                      "note": "measured wave-instructions by class (SQ_INSTS_VALU_*_F64 at 4 cycles, every other "
                              "VALU instruction at 2) / (1024 SIMDs x 2.4 GHz x kernel time)"})
        # the f64 flops the kernel really issued (adds and multiplies 1, FMAs 2, per lane of 64)
        meas = (f64[0] + f64[1] + 2 * f64[2]) * 64 * per_sub / (surf_us * 1e-6) / 1e12
    else:
        cyc = valu * VALU_CYCLES_F64
        issue.update({"issue_cycles_per_sub_timestep": cyc, "frac": cyc / (surf_us * 1e-6) / SIMD_CYCLES_PER_SEC,
                      "note": "no per-class counts in the committed file: every VALU instruction booked at 4 cycles "
                              "(an upper bound of the occupancy)"})
    return issue, meas


def hbm_roofline(ab, surf_us, substep_us, n, counters, src, stale, kernel, latency_bound=False):
    """Streamed march against the HBM roofline (algorithmic bytes / kernel time). latency_bound: the
    workload moves a tenth of the bytes the chip could (BASELINE config 5: 0.1 GB per sub-timestep of windows whose
    no-mass loop re-evaluates the cavity's Nusselt correlations every pass, surface.rs:814) — the line then names the
    instruction issue of the dependent chains as its bound and keeps the HBM figures beside it."""
    achieved = ab / (surf_us * 1e-6) / 1e9
    traffic = counters.get("hbm_traffic_bytes_per_launch") if (counters and not stale) else None
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_source": src, "kernel": kernel,
         "algorithmic_bytes_per_launch": ab, "kernel_us": surf_us, "substep_us": substep_us,
         "frac_whole_sub_timestep": ab / (substep_us * 1e-6) / 1e9 / HBM_PEAK_GBS if substep_us else None,
         "samples": n}
    if counters:
        r["counters_stale"] = bool(stale)
        if not stale:
            for k in ("SQ_WAIT_ANY_frac_of_wave_cycles", "SQ_ACTIVE_INST_VALU_frac_of_wave_cycles", "valu_insts_per_launch"):
                if k in counters:
                    r[k] = counters[k]
            r["issue"], meas = issue_view(counters, 1.0 / counters["workload"].get("substeps_per_launch", 1), surf_us)
            if meas is not None:
                r["measured_f64_tflops"] = meas
            if traffic:
                r["hbm_gbs_measured_traffic"] = traffic / (surf_us * 1e-6) / 1e9
            ks = counters.get("kernel", "")
            rk = rocprof_kernel_us(src, ks.split("<")[0] + ("<" + ks.split("<")[1] if "<" in ks else ""))
            if rk:
                r["kernel_us_rocprof"] = rk
                r["frac_rocprof"] = ab / (rk * 1e-6) / 1e9 / HBM_PEAK_GBS
                r["kernel_us_note"] = ("kernel_us: HIP events around the launches of a sub-timestep, live (they include the gap "
                                       "between its kernels when launched outside a graph); kernel_us_rocprof: sum of the "
                                       "kernels' mean durations in the committed rocprofv3 --kernel-trace --stats of this "
                                       "workload, same kernel sources")
    if latency_bound:
        r["hbm_frac"] = r["frac"]
        r["bound_note"] = ("not an HBM-bound workload: %.2f GB of algorithmic bytes per sub-timestep; what bounds it is the "
                           "latency of dependent f64 chains (Cavity::u_value and the Nusselt correlations inside the "
                           "no-mass loop, two wavefronts per SIMD)" % (ab / 1e9))
        if r.get("issue") and r["issue"].get("frac") is not None:
            r.update({"bound": "valu_issue", "achieved": r["issue"]["issue_cycles_per_sub_timestep"] / (surf_us * 1e-6) / 1e9,
                      "peak": SIMD_CYCLES_PER_SEC / 1e9, "unit": "Gcycle/s", "frac": r["issue"]["frac"],
                      "hbm_achieved_gbs": achieved})
    return r


def fused_roofline(nodes_local, n_nodes_counted, P, surf_us, substep_us, n_samples, algorithmic_bytes, counters, src, stale):
    """Cluster-resident march: the state is re-used on chip (1 GB of HBM traffic for 23.5 GB of algorithmic bytes at
    20 sub-timesteps), so neither HBM nor MFMA bounds it: the line is the ALGORITHMIC f64 work against the f64 vector
    peak. `issue` is the instruction-issue view from the committed counters of this workload (per-class counts)."""
    useful = ALGORITHMIC_FLOPS_PER_NODE_UPDATE * nodes_local / (surf_us * 1e-6) / 1e12
    rl = {"bound": "valu_f64", "achieved": useful, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
          "frac": useful / F64_PEAK_TFLOPS, "useful_f64_frac": useful / F64_PEAK_TFLOPS,
          "algorithmic_flops_per_node_update": ALGORITHMIC_FLOPS_PER_NODE_UPDATE,
          "peak_note": "256 CUs x 4 SIMDs x 16 f64 lanes x 2 flops x 2.4 GHz",
          "kernel": "k_surfaces_fast<M,...,FUSED> (cluster-resident march: %d sub-timesteps of iterate_surfaces + zone "
                    "balance per launch, node temperatures in registers)" % P,
          "sub_timesteps_per_launch": P, "kernel_us": surf_us * P, "kernel_us_per_sub_timestep": surf_us,
          "substep_us": substep_us, "samples": n_samples, "traffic": None, "counters_source": src,
          "counters_stale": bool(stale) if counters else None,
          "equivalent_streaming_gbs": algorithmic_bytes / (surf_us * 1e-6) / 1e9,
          "equivalent_streaming_note": "algorithmic (streaming) bytes of the same sub-timesteps / kernel time: what a "
                                       "streamed march would have to sustain; not a roofline fraction"}
    if counters and not stale:
        scale = nodes_local / counters["workload"]["nodes_total"]  # (a rank of a sharded run launches its share)
        per_sub = scale / counters["workload"]["substeps_per_launch"]
        issue, meas = issue_view(counters, per_sub, surf_us)
        if meas is not None:
            rl["measured_f64_tflops"] = meas
            rl["measured_f64_frac"] = meas / F64_PEAK_TFLOPS
        rl["issue"] = issue
        t = counters.get("hbm_traffic_bytes_per_launch")
        if t:
            rl["traffic"] = t * scale
            rl["hbm_gbs_measured_traffic"] = rl["traffic"] / (surf_us * P * 1e-6) / 1e9
    else:
        rl["issue"] = None
    return rl


def streaming_leg(md, state, args, config, dt, steps=60, warmup=10):
    """The per-sub-timestep kernels on their own: a batch planned without the cluster-resident march, HIP events
    around every sub-timestep."""
    from heat_amd import HeatBatch, modeldict as mdl
    with HeatBatch(md, nodes_per_lane=args.nodes_per_lane, no_palette=args.no_palette, no_fusion=True) as b:
        b.upload_state(state)
        if args.settle_ms > 0:
            settle(b.march_resident, mdl.weather_series(20, dt), b.synchronize, args.settle_ms)
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
    counters, src, stale = committed_counters(config, int(md["n_surfaces"]), int(md["node_offset"][-1]), "streamed")
    r = hbm_roofline(ab, surf_us, substep_us, n, counters, src, stale,
                     "streamed march: iterate_surfaces, one sub-timestep per launch, state streamed through HBM")
    r["node_updates_per_sec"] = int(md["node_offset"][-1]) * steps / el
    r["kernel_classes[M4,M8,M16,small,general]"] = counts
    return r


def torch_device_sync(local_rank):
    """(torch is plumbing here: its device-wide synchronize brackets the timed region as the contract words it)"""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
            return torch.cuda.synchronize
    except Exception:
        pass
    return None


def single_gpu_leg(config, args, K, W, P, seed, local_rank, main, cpu_seconds):
    """One workload on one GPU: W untimed + K timed sub-timesteps as march calls of P, the kernel's roofline from HIP
    events inside the timed region, the CPU baseline on a bounded sample. main: the headline's extras too."""
    from heat_amd import HeatBatch, modeldict as mdl
    md, state, workload = build_config(config, args, 45.0, seed)
    dt = float(md["dt"])
    n_nodes = int(md["node_offset"][-1])
    batch = HeatBatch(md, device=local_rank, nodes_per_lane=args.nodes_per_lane, use_graph=True,
                      no_palette=args.no_palette, no_fusion=args.no_fusion)
    try:
        batch.upload_state(state)
        torch_sync = torch_device_sync(local_rank)

        def barrier():
            batch.synchronize()  # the batch's own streams; reports device-side numerical flags
            if torch_sync:
                torch_sync()

        settle_calls, settle_s = settle(batch.march_resident, mdl.weather_series(P, dt), barrier, args.settle_ms) if args.settle_ms > 0 else (0, 0.0)
        if W > 0:
            march_in_calls(batch.march_resident, mdl.weather_series(W, dt), P)
        barrier()
        # HIP events around the kernels inside the timed region: every call of a cluster-resident march (two events per
        # launch); of a streamed march ONE call in three, or the first call only when the region has fewer (three events
        # per sub-timestep, issued eagerly — the other calls replay the hipGraph as an untimed march does)
        n_local = batch.n_surfaces_in_batch
        fused = (not args.no_fusion) and batch.n_fused_surfaces > 0 and P >= (1 if n_local <= 8192 else 2)
        batch.set_timing(0 if args.no_timing else (1 if fused else max(3, -(-K // P))))
        weather_k = mdl.weather_series(K, dt, t0=dt * W)
        t0 = time.perf_counter()
        march_in_calls(batch.march_resident, weather_k, P)
        barrier()
        elapsed = time.perf_counter() - t0
        surf_us, substep_us, n_samples = batch.get_timing()
        batch.set_timing(False)
        ab = batch.algorithmic_bytes
        counts = batch.class_counts()
        n_fused = batch.n_fused_surfaces if not args.no_fusion else 0
        res = {"value": n_nodes * K / elapsed, "unit": "node-updates/s", "steps": K, "warmup": W,
               "ms_per_step": elapsed / K * 1e3, "sub_timesteps_per_sec": K / elapsed,
               "settle": {"sub_timesteps": settle_calls * P, "ms": settle_s * 1e3, "note": SETTLE_NOTE},
               "config": {"workload": workload + "; one step = one sub-timestep (iterate_surfaces + zone update)",
                          "config": config, "surfaces": int(md["n_surfaces"]), "nodes": n_nodes,
                          "zones": int(md["n_zones"]), "dt_s": dt, "substeps_per_march": P,
                          "surfaces_on_rank0": n_local, "kernel_classes_rank0[M4,M8,M16,small,general]": counts,
                          "surfaces_in_cluster_resident_march_rank0": n_fused, "n_shared_zones": 0,
                          "parallelism": "single GPU"}}
        if n_samples > 0 and fused:
            counters, src, stale = committed_counters(config, int(md["n_surfaces"]), n_nodes, "fused", P)
            res["roofline"] = fused_roofline(batch.n_nodes, n_nodes, P, surf_us, substep_us, n_samples, ab, counters, src, stale)
        elif n_samples > 0:
            counters, src, stale = committed_counters(config, int(md["n_surfaces"]), n_nodes, "streamed")
            res["roofline"] = hbm_roofline(ab, surf_us, substep_us, n_samples, counters, src, stale,
                                           "streamed march: k_surfaces_stream / k_surfaces_fast / k_surfaces_small "
                                           "(iterate_surfaces: RK4 stencil + boundary updates, one sub-timestep per launch)",
                                           latency_bound=config in ("5", "5x10"))
            # the same bytes over the wall-clock time of a step of the timed region (graph replay, launches and all)
            res["roofline"]["frac_wall_clock_step"] = ab / (elapsed / K) / 1e9 / HBM_PEAK_GBS
        if main and not args.no_extras:
            # the drop-in call on a caller-owned host state (PCIe-inclusive; never `value`)
            st = state.copy()
            wcall = mdl.weather_series(P, dt)
            batch.march(st, wcall)
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                batch.march(st, wcall)
            tc = (time.perf_counter() - t0) / reps
            res["caller_owned"] = {"value": n_nodes * P / tc, "unit": "node-updates/s", "ms_per_call": tc * 1e3,
                                   "sub_timesteps_per_call": P, "state_megabytes": st.nbytes / 1e6,
                                   "note": "heat_batch_march: inputs up, march, outputs down, caller's pageable numpy array"}
            t0 = time.perf_counter()
            for _ in range(reps):
                batch.march(st, wcall, outputs=HeatBatch.OUT_SCALARS | HeatBatch.OUT_ZONES)
            tc = (time.perf_counter() - t0) / reps
            res["caller_owned"]["scalars_and_zones_only"] = {
                "value": n_nodes * P / tc, "ms_per_call": tc * 1e3,
                "note": "heat_batch_march_ex(HEAT_OUT_SURFACE_SCALARS | HEAT_OUT_ZONE_TEMPERATURES): what the Rust shim "
                        "asks for every call; node temperatures on demand"}
            # how the sub-timesteps per march call change the picture (the reference's config 1 runs 2 per call)
            sens = {}
            if args.settle_ms > 0:  # (the copies above left the chip idle)
                settle(batch.march_resident, mdl.weather_series(P, dt), batch.synchronize, args.settle_ms)
            for p in (2, 4, 5, 20):
                wv = mdl.weather_series(p, dt)
                calls = max(4, 80 // p)
                batch.march_resident(wv)
                batch.synchronize()
                t0 = time.perf_counter()
                for _ in range(calls):
                    batch.march_resident(wv)
                batch.synchronize()
                sens[str(p)] = n_nodes * p * calls / (time.perf_counter() - t0)
            res["value_by_substeps_per_march"] = sens
    finally:
        batch.close()
    if fused and not args.no_streaming_leg:
        res["roofline_streaming"] = streaming_leg(md, state, args, config, dt, steps=60 if main else 20)
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(md, state, dt, cpu_seconds)
    return res


def sharded_leg(config, args, K, W, P, seed, rank, world, local_rank, weak):
    """One workload on `world` ranks (one process per GPU). Returns the rank-0 view; every rank must call it."""
    import torch
    import torch.distributed as dist
    from heat_amd import modeldict as mdl
    from heat_amd.sharded import ShardedMarch, partition_model
    if weak:
        md, state = build_weak_shard(args.surfaces, args.nodes, max(1, args.surfaces // 100), rank, world, 45.0, seed)
        workload = ("north_star headline, weak scaling: %d all-massive surfaces x %d nodes PER GPU, zones of 100 walls "
                    "offset by half a zone against the rank boundaries, dt = 45 s" % (args.surfaces, args.nodes))
    else:
        md, state, workload = build_config(config, args, 45.0, seed)
    dt = float(md["dt"])
    n_nodes_model = int(md["node_offset"][-1])
    forced = None
    if args.force_shared_zones > 0:
        forced = np.unique(np.linspace(0, int(md["n_zones"]) - 1, args.force_shared_zones).astype(np.int32))
    ranks = n_shared_partition = None
    if not weak:
        # every rank cuts the same model the same way (host-only, deterministic): whole clusters per rank
        ranks, n_shared_partition = partition_model(md, world)
    dev = local_rank % max(torch.cuda.device_count(), 1)
    sm = ShardedMarch(md, rank, world, device_index=dev, collective=args.collective, force_shared=forced,
                      rank_of_surface=ranks, n_shared_in_partition=n_shared_partition,
                      nodes_per_lane=args.nodes_per_lane, no_palette=args.no_palette, no_fusion=args.no_fusion,
                      use_graph=True)
    try:
        batch = sm.batch
        batch.upload_state(state)

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()
            batch.synchronize()  # reports device-side numerical flags

        settle_calls, settle_s = 0, 0.0
        if args.settle_ms > 0:
            # (every rank the same number of calls — the marches may hold a collective —: blocks of calls whose length
            # every rank derives from the same all-reduced time of the block before)
            wsettle = mdl.weather_series(P, dt)
            n_block = 1
            while settle_s < args.settle_ms * 1e-3 and settle_calls < 4000:
                _, t_block = settle(sm.march_resident, wsettle, barrier, 0.0, n_calls=n_block)
                tt = torch.tensor([t_block], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t_block = float(tt.item())
                settle_s += t_block
                settle_calls += n_block
                left = args.settle_ms * 1e-3 - settle_s
                n_block = int(min(500, max(1, left / max(t_block / n_block, 1e-6))))
        if W > 0:
            march_in_calls(sm.march_resident, mdl.weather_series(W, dt), P)
        barrier()
        n_local = batch.n_surfaces_in_batch
        fused = (not args.no_fusion) and batch.n_fused_surfaces > 0 and P >= (1 if n_local <= 8192 else 2)
        batch.set_timing(0 if args.no_timing else (1 if fused else max(3, -(-K // P))))
        t0 = time.perf_counter()
        march_in_calls(sm.march_resident, mdl.weather_series(K, dt, t0=dt * W), P)
        barrier()
        elapsed = time.perf_counter() - t0
        surf_us, substep_us, n_samples = batch.get_timing()
        batch.set_timing(False)
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        total_nodes = n_nodes_model * (world if weak else 1)
        n_shared = batch.n_shared_zones
        how = {"native": "library-owned RCCL communicator, all-gather in the batch's one stream",
               "torch": "torch.distributed.all_gather_into_tensor", "none": "no collective issued"}[sm.collective]
        res = {"value": total_nodes * K / elapsed, "unit": "node-updates/s", "steps": K, "warmup": W,
               "ms_per_step": elapsed / K * 1e3, "sub_timesteps_per_sec": K / elapsed,
               "settle": {"sub_timesteps": settle_calls * P, "ms": settle_s * 1e3, "note": SETTLE_NOTE},
               "n_shared_zones": n_shared, "collective": sm.collective, "comm_ranks": sm.comm_ranks,
               "collective_fallback": sm.collective_fallback,
               "config": {"workload": workload + "; one step = one sub-timestep (iterate_surfaces + zone update)",
                          "config": config, "surfaces": int(md["n_surfaces"]) * (world if weak else 1),
                          "nodes": total_nodes, "zones": int(md["n_zones"]), "dt_s": dt, "substeps_per_march": P,
                          "surfaces_on_rank0": n_local,
                          "kernel_classes_rank0[M4,M8,M16,small,general]": batch.class_counts(),
                          "surfaces_in_cluster_resident_march_rank0": batch.n_fused_surfaces if not args.no_fusion else 0,
                          "n_shared_zones": n_shared,
                          "parallelism": "surfaces sharded %d-way %s, zones replicated; %d zones shared between ranks (%s)" % (
                              world, "by rank-local models" if weak else "along the zone-connected clusters (heat_partition)",
                              n_shared, how)}}
        ab = batch.algorithmic_bytes
        if n_samples > 0 and fused:
            counters, src, stale = committed_counters(config, int(md["n_surfaces"]), n_nodes_model, "fused", P)
            res["roofline"] = fused_roofline(batch.n_nodes, n_nodes_model, P, surf_us, substep_us, n_samples, ab, counters, src, stale)
        elif n_samples > 0:
            res["roofline"] = hbm_roofline(ab, surf_us, substep_us, n_samples, None, None, None,
                                           "streamed march of rank 0's shard (iterate_surfaces, one sub-timestep per launch)")
        return res
    finally:
        sm.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=CONFIGS, default="headline",
                    help="workload of the main leg: the north_star headline (default) or a BASELINE.json config")
    ap.add_argument("--no-configs", action="store_true",
                    help="headline only: skip the legs of BASELINE configs 3, 5, 2 and partitions")
    ap.add_argument("--config-steps", type=int, default=60, help="timed sub-timesteps of each of those legs")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: shard the same model (BASELINE config 4; default) or give every rank its own")
    ap.add_argument("--surfaces", type=int, default=1_000_000, help="surfaces of the model (weak scaling: per GPU)")
    ap.add_argument("--nodes", type=int, default=32)
    ap.add_argument("--nodes-per-lane", type=int, default=0)
    ap.add_argument("--no-palette", action="store_true", help="keep dt/mass and U as per-node arrays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--substeps-per-march", type=int, default=20,
                    help="sub-timesteps per march call (ThermalModel::march runs dt_subdivisions of them per call)")
    ap.add_argument("--no-fusion", action="store_true",
                    help="stream every sub-timestep through HBM (no cluster-resident march)")
    ap.add_argument("--no-streaming-leg", action="store_true", help="skip the second leg that measures the streamed kernel")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the caller-owned-state and sub-timesteps-per-call measurements")
    ap.add_argument("--no-timing", action="store_true", help="do not record HIP events in the timed region")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed march calls before the warmup steps, this long, so that the timed region runs at the "
                         "clock the chip holds under load (0: none)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="drive the multi-GPU code path (ShardedMarch + zone exchange) even with one rank")
    ap.add_argument("--force-shared-zones", type=int, default=0,
                    help="single-GPU rehearsal of the exchange: declare this many zones shared (implies --force-sharded)")
    ap.add_argument("--no-exchange-leg", action="store_true", help="N > 1: skip the config-3 leg that exercises the collective")
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
    P = max(1, args.substeps_per_march)
    seed = 20260401
    sharded = world > 1 or args.force_sharded or args.force_shared_zones > 0
    weak = args.scaling == "weak" and sharded
    if weak and args.config != "headline":
        raise SystemExit("--scaling weak exists for the headline workload only")

    result = {"metric": "surface-node-updates/sec", "value": None, "unit": "node-updates/s", "n_gpus": world,
              "steps": K, "warmup": W, "ms_per_step": None, "higher_is_better": True,
              "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
              "kernel_sources_sha256": kernel_source_hash()[:16]}
    try:
        if sharded:
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29511")
                # (HEAT_AMD_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow with several ranks on ONE GPU, where
                # RCCL refuses a second rank on the same device; only barriers and the timing reduction go through it)
                backend = os.environ.get("HEAT_AMD_BENCH_BACKEND", "nccl")
                if backend == "nccl":
                    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
                else:
                    dist.init_process_group(backend, rank=rank, world_size=world)
            result.update(sharded_leg(args.config, args, K, W, P, seed, rank, world, local_rank, weak))
            if world > 1 and not weak and not args.no_exchange_leg and args.config == "headline":
                # the collective at work: config 3's ring of zones is one cluster, cut by surface ranges
                try:
                    leg = sharded_leg("3", args, args.config_steps, 20, P, seed, rank, world, local_rank, False)
                except Exception as e:  # noqa: BLE001 — the main line still goes out
                    leg = {"error": "%s: %s" % (type(e).__name__, e)}
                result["exchange_leg"] = leg
            dist.destroy_process_group()
        else:
            result.update(single_gpu_leg(args.config, args, K, W, P, seed, local_rank, True, 12.0))
            if args.config == "headline" and not args.no_configs:
                result["configs"] = {}
                for name in ("3", "5", "2", "partitions", "buildings40", "5x10"):
                    t0 = time.perf_counter()
                    try:
                        leg = single_gpu_leg(name, args, args.config_steps, 20, P, seed, local_rank, False, 4.0)
                    except Exception as e:  # noqa: BLE001 — a failing leg must not take the line with it
                        leg = {"error": "%s: %s" % (type(e).__name__, e)}
                    leg["leg_seconds"] = time.perf_counter() - t0
                    result["configs"][name] = leg
    except Exception as e:  # noqa: BLE001 — the line goes out whatever happened, and says what did
        traceback.print_exc(file=sys.stderr)
        result["error"] = "%s: %s" % (type(e).__name__, e)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if "error" in result:
        sys.exit(1)


if __name__ == "__main__":
    main()
