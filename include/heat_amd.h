/*
 * heat_amd.h — C ABI of the MI355X-native wall heat-conduction path.
 *
 * This is the drop-in boundary for the hot path of SIMPLE-BuildingSimulation/heat
 * (reference paths below are relative to the reference repository root):
 *
 *   ThermalModel::new + allocate_memory   src/model.rs:193-354   -> heat_batch_create
 *   ThermalModel::march (sub-dt loop)      src/model.rs:359-427   -> heat_batch_march
 *   iterate_surfaces                       src/model.rs:102-180   -> (inside march) heat_batch_step_surfaces
 *   calculate_zones_abc + estimate_zones_future_temperatures
 *                                          src/model.rs:489-597,650-674 -> heat_batch_step_zones
 *   SurfaceTrait::{get,set}_node_temperatures & the scalar slot accessors
 *                                          src/surface_trait.rs:81-164 -> heat_batch_upload_state / _download_state
 *
 * The caller (a Rust shim implementing `SimulationModel for GpuThermalModel`,
 * see INTEGRATION.md) keeps owning the flat `SimulationState` array; this
 * library owns a device-resident mirror of the slots the path touches, laid out
 * as a lane-blocked structure of arrays in HBM.
 *
 * Plain C types only. Every function returns 0 on success, a negative
 * HEAT_E_* code for an invalid call/descriptor, or a positive HEAT_N_* code
 * for a numerical failure detected on the device (the reference panics there).
 * heat_last_error() returns a human-readable message for the last failure on
 * the calling thread. The library never aborts the process.
 */
#ifndef HEAT_AMD_H
#define HEAT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEAT_AMD_ABI_VERSION 1

/* simple_model::Boundary as consumed at src/surface.rs:611-702 */
enum heat_boundary_kind {
    HEAT_BOUNDARY_SPACE = 0,
    HEAT_BOUNDARY_AMBIENT = 1, /* Boundary::AmbientTemperature { temperature } */
    HEAT_BOUNDARY_OUTDOOR = 2,
    HEAT_BOUNDARY_GROUND = 3   /* rejected: the reference panics (surface.rs:642,687; model.rs:92) */
};

/* src/gas.rs:45-74 */
enum heat_gas { HEAT_GAS_AIR = 0, HEAT_GAS_ARGON = 1, HEAT_GAS_KRYPTON = 2, HEAT_GAS_XENON = 3 };

enum heat_status {
    HEAT_OK = 0,
    /* invalid call / descriptor (reference: Err(String) or a setup-time panic) */
    HEAT_E_INVALID_ARG = -1,
    HEAT_E_GROUND_BOUNDARY = -2, /* surface.rs:642,687 */
    HEAT_E_UVALUE_NONE = -3,     /* discretization.rs:53 */
    HEAT_E_SIZE = -4,            /* slot or index out of range */
    HEAT_E_DEVICE = -5,          /* HIP runtime failure (message has the HIP error string) */
    HEAT_E_TOO_MANY_NODES = -6,
    HEAT_E_COMM = -7,            /* RCCL not loadable, or a collective failed (message has RCCL's error string) */
    /* numerical failure on the device (reference: assert!/unreachable! panics) */
    HEAT_N_NAN_HS = 1,           /* surface.rs:704-707 */
    HEAT_N_NAN_NOMASS = 2,       /* surface.rs:850 */
    HEAT_N_NAN_ZONE = 3,         /* model.rs:417-420 */
    HEAT_N_UNREACHABLE = 4       /* convection.rs:104, gas.rs:219,296 */
};

/* src/cavity.rs:28-50 */
typedef struct heat_cavity {
    double thickness;
    double height;
    double angle; /* radians; 0 horizontal, pi/2 vertical */
    double eout;
    double ein;
    int32_t gas;  /* enum heat_gas */
    int32_t reserved;
} heat_cavity;

/* Weather of one sub-timestep (model.rs:371-382). The shim converts degrees to radians
 * (`wind_direction.to_radians()`, model.rs:373). */
typedef struct heat_weather {
    double dry_bulb;       /* C */
    double wind_direction; /* radians */
    double wind_speed;     /* m/s */
} heat_weather;

/*
 * Everything ThermalModel::new derives and the hot path reads, flattened.
 * Surfaces first, then fenestrations (the reference iterates them in that
 * order, model.rs:388-408). All arrays are host memory, copied by
 * heat_batch_create; they need not outlive the call.
 */
typedef struct heat_batch_desc {
    int32_t abi_version; /* HEAT_AMD_ABI_VERSION */
    int32_t reserved;
    int64_t n_surfaces;
    int64_t n_zones;
    int64_t n_cavities;
    int64_t n_state; /* length of the caller's SimulationState array */
    double dt;       /* ThermalModel::dt, model.rs:76,326-330 */

    /* Discretization::segments, CSR over surfaces (discretization.rs:73) */
    const int64_t *node_offset; /* [n_surfaces+1] */
    const double *mass;         /* segments[i].0 ; a node is massive iff mass >= 1e-5 (discretization.rs:149) */
    const double *uvalue;       /* UValue::Solid(u) -> u ; UValue::Back -> 0 ; NaN = UValue::None (rejected) */
    const int32_t *seg_cavity;  /* UValue::Cavity -> index into cavities, else -1 ; NULL when n_cavities == 0 */
    const double *front_alpha;  /* ThermalSurfaceData::front_alphas, surface.rs:366 */
    const double *back_alpha;   /* ThermalSurfaceData::back_alphas, surface.rs:370 */
    const heat_cavity *cavities;

    /* ThermalSurfaceData fields, surface.rs:315-381 */
    const int32_t *front_kind, *back_kind; /* enum heat_boundary_kind */
    const int32_t *front_zone, *back_zone; /* front/back_space_index (used when kind == SPACE) */
    const double *front_ambient, *back_ambient; /* used when kind == AMBIENT */
    const double *front_emissivity, *back_emissivity;
    const double *area, *perimeter;
    const double *cos_tilt;
    const double *normal_x, *normal_y;
    const double *wind_modifier;
    const double *front_hs_fix, *back_hs_fix; /* debug-only overrides, surface.rs:374-380; NULL or NaN = none */

    /* SimulationState slots (surface.rs:428-442; surface_trait.rs:223-378) */
    const int64_t *first_node_slot; /* node slots of a surface are contiguous */
    const int64_t *hs_front_slot, *hs_back_slot;
    const int64_t *flow_front_slot, *flow_back_slot;
    const int64_t *solar_front_slot, *solar_back_slot;
    const int64_t *ir_front_slot, *ir_back_slot;

    /* ThermalZone (zone.rs:28-56) */
    const double *zone_volume;
    const int64_t *zone_slot; /* Space dry-bulb temperature slot */
} heat_batch_desc;

typedef struct heat_batch heat_batch;

/* Options for heat_batch_create_ex. Zero-initialise for defaults. */
typedef struct heat_batch_options {
    int32_t device;          /* HIP device ordinal; -1 = current device */
    int32_t force_general;   /* 1: route every surface through the general (catch-all) kernel */
    int32_t nodes_per_lane;  /* 0 = auto; else 4, 8 or 16 (fast-path blocking factor) */
    int32_t use_graph;       /* 1: replay the sub-timestep as a hipGraph inside heat_batch_march */
    void *stream;            /* hipStream_t to run on; NULL = a stream owned by the batch */
    /* Multi-GPU (one process per GPU): this rank holds a shard of the surfaces but all zones.
     * With n_ranks > 1 the caller either gives the batch a communicator (heat_batch_comm_init, below) and
     * marches as on one GPU, or brings its own collective and alternates
     * heat_batch_step_surfaces -> all-gather of heat_batch_zone_partials -> heat_batch_step_zones. */
    int32_t n_ranks;
    int32_t rank;
    int32_t no_palette;      /* 1: keep dt/mass and U as per-node arrays even where a palette would do */
    int32_t no_fusion;       /* cluster-resident march (see heat_batch_set_fusion): 0 = plan it for the clusters the cost
                              * model expects to gain, 1 = never, 2 = for every cluster that structurally can (tests) */
} heat_batch_options;

/* ≙ ThermalModel::new + allocate_memory: validates, packs and uploads the constants. */
int heat_batch_create(const heat_batch_desc *desc, heat_batch **out);
int heat_batch_create_ex(const heat_batch_desc *desc, const heat_batch_options *opt, heat_batch **out);
void heat_batch_destroy(heat_batch *b);

/* Copies every slot the path touches (node temperatures, hs, flows, irradiances, zone
 * dry-bulb) from / to the caller's SimulationState. */
int heat_batch_upload_state(heat_batch *b, const double *state, size_t n_state);
int heat_batch_download_state(heat_batch *b, double *state, size_t n_state);
/* Only what other modules write between two march calls: solar + IR irradiance slots
 * and zone dry-bulb temperatures. The zone slots are taken only while `state` holds what this path last computed
 * for them: after a march whose outputs left HEAT_OUT_ZONE_TEMPERATURES out (or a resident march without a download)
 * the caller's zone slots are older than the device's and are NOT read — heat_batch_upload_state takes everything. */
int heat_batch_upload_inputs(heat_batch *b, const double *state, size_t n_state);

/*
 * ≙ ThermalModel::march (model.rs:359-427): n_sub sub-timesteps.
 * zone_a0 / zone_b0 (nullable, [n_zones]) are the terms of calculate_zones_abc that do
 * not come from surfaces (HVAC, luminaires, infiltration, ventilation; model.rs:500-544),
 * evaluated by the caller.
 * Uploads the inputs from `state`, marches, downloads the outputs into `state`.
 */
int heat_batch_march(heat_batch *b, double *state, size_t n_state, const heat_weather *weather,
                     int32_t n_sub, const double *zone_a0, const double *zone_b0);
/* Data at this boundary: the call uploads only the slots other modules write between two marches
 * — the 4 S irradiance slots and the zones' dry-bulb slots, gathered on the host into pinned memory, one 32 MB copy
 * per million surfaces — and downloads only the outputs of this module, through two pinned staging halves with the
 * host scatter on a thread pool (HEAT_AMD_HOST_THREADS, default min(16, cores)) overlapping the copies.
 * heat_batch_march_ex chooses which outputs come back every call: a caller that reads the node temperatures only
 * now and then (they are 8 n of every surface's 8 n + 32 output bytes) leaves HEAT_OUT_NODE_TEMPERATURES out and
 * fetches them with heat_batch_download_outputs when needed; the device-resident state is always complete. */
enum heat_outputs {
    HEAT_OUT_NODE_TEMPERATURES = 1, /* SurfaceTrait::set_node_temperatures, surface_trait.rs:107-125 */
    HEAT_OUT_SURFACE_SCALARS = 2,   /* hs front / back, convective heat flow front / back (model.rs:154-169) */
    HEAT_OUT_ZONE_TEMPERATURES = 4, /* dry-bulb temperature of the zones this batch owns (model.rs:410-423) */
    HEAT_OUT_ALL = 7
};
int heat_batch_march_ex(heat_batch *b, double *state, size_t n_state, const heat_weather *weather, int32_t n_sub,
                        const double *zone_a0, const double *zone_b0, int32_t what);
int heat_batch_download_outputs(heat_batch *b, double *state, size_t n_state, int32_t what);

/* Same, but on the device-resident state only (no host traffic; asynchronous on the batch's
 * stream until heat_batch_synchronize / a download). */
int heat_batch_march_resident(heat_batch *b, const heat_weather *weather, int32_t n_sub,
                              const double *zone_a0, const double *zone_b0);
int heat_batch_synchronize(heat_batch *b); /* waits, then reports device-side numerical flags */
/* Where the numerical failure heat_batch_synchronize / heat_batch_march last reported was seen FIRST (the reference's
 * panics name the offending values, surface.rs:704-707; model.rs:417-420): *index = the surface's number in the
 * descriptor — or the zone's, when *kind == HEAT_N_NAN_ZONE found by the zone balance itself (the cluster-resident
 * march reports a surface of the zone's cluster instead) — and *kind = the HEAT_N_* code seen there. -1 / 0 when no
 * failure has been reported yet. Host-side bookkeeping: no device access. */
int heat_batch_failed_surface(const heat_batch *b, int64_t *index, int32_t *kind);

/* Split-phase sub-timestep, for the sharded (multi-GPU) case. All asynchronous on the stream.
 * step_surfaces ≙ iterate_surfaces over this rank's surfaces + this rank's partial (a,b) sums.
 * step_zones    ≙ the zone update from `gathered` = n_ranks consecutive partial blocks
 *                 (device pointer, layout [rank][2][n_zones]: a then b), summed in rank order. */
int heat_batch_set_weather(heat_batch *b, const heat_weather *weather, int32_t n_sub,
                           const double *zone_a0, const double *zone_b0);
int heat_batch_step_surfaces(heat_batch *b, int32_t sub_step);
int heat_batch_step_zones(heat_batch *b, const double *gathered_dev, int32_t n_blocks);
double *heat_batch_zone_partials(heat_batch *b); /* device pointer, [2][n_zones] doubles */
/* Makes step_surfaces write the partial sums into caller-owned device memory ([2][n_zones] doubles,
 * e.g. a tensor the caller hands to its collective); NULL restores the batch's own buffer. */
int heat_batch_use_partials(heat_batch *b, double *partials_dev);
/* Sharded batches, compact exchange. heat_batch_touched_zones fills mask[n_zones] with 1 for every zone a surface
 * of this batch faces. After the ranks have agreed on the zones more than one of them touches,
 * heat_batch_set_shared_zones(shared_zone[n_shared]: global zone numbers, the same list in the same order on every
 * rank) switches the split-phase sequence to the compact form: heat_batch_step_surfaces updates the zones only this
 * rank touches at once and writes the partial (a, b) of the shared ones into the partials buffer, laid out
 * [2][n_shared]; heat_batch_step_zones(gathered, n_blocks) then updates the shared zones from the gathered blocks
 * [block][2][n_shared], summed in block order. Zones this rank does not touch are not kept up to date on it. */
int heat_batch_touched_zones(const heat_batch *b, uint8_t *mask);
int heat_batch_set_shared_zones(heat_batch *b, const int32_t *shared_zone, int32_t n_shared);

/*
 * Library-owned collective (the default multi-GPU mode; one process per GPU): the batch holds an RCCL
 * communicator and heat_batch_march_resident / heat_batch_march run the whole sharded sub-timestep on the batch's
 * stream: surfaces -> zones only this rank touches + partial (a, b) of the shared ones -> ncclAllGather of the
 * [2][n_shared] blocks over xGMI -> shared zones updated from the blocks summed in rank order. No torch, no second
 * stream: a kernel, a collective and a kernel in one queue.
 *   heat_comm_unique_id   ncclGetUniqueId; one rank calls it and the host program hands the bytes to every rank
 *                         (any means: MPI, a file, torch.distributed's store, ...).
 *   heat_batch_comm_init  ncclCommInitRank(n_ranks, rank of the batch's options) — collective: every rank calls it;
 *                         then the ranks agree on the shared zones (an all-reduce of the touched masks) and the
 *                         batch is switched to the compact exchange (as heat_batch_set_shared_zones does).
 * RCCL is loaded at run time (dlopen "librccl.so.1"); without it both calls return HEAT_E_COMM.
 *   heat_comm_available   HEAT_OK when RCCL can be loaded (no collective inside: the ranks can vote on it BEFORE any
 *                         of them enters the collective heat_batch_comm_init).
 *   heat_batch_comm_init_ex  as heat_batch_comm_init, with extra_shared[n_extra] zones exchanged as well (the union
 *                         with the agreed list; tests and single-GPU rehearsals of the exchange).
 * Zones NO rank faces still follow their a0 / b0 terms (model.rs:410-423): rank z % n_ranks finishes zone z
 * (heat_batch_comm_init and heat_batch_create_shard arrange that; heat_batch_set_owned_zones for callers that cut
 * their shards themselves: owned[n_zones], OR-ed with the zones the batch's surfaces face).
 * A sharded batch that shares no zone with another rank — a partition along the clusters, heat_partition —
 * needs no communicator at all: after heat_batch_set_shared_zones(b, NULL, 0) (heat_batch_create_shard does it)
 * heat_batch_march[_resident] run as on a single GPU, on the zones the batch owns.
 */
#define HEAT_COMM_ID_BYTES 128
int heat_comm_available(void);
int heat_comm_unique_id(uint8_t id[HEAT_COMM_ID_BYTES]);
int heat_batch_comm_init(heat_batch *b, const uint8_t id[HEAT_COMM_ID_BYTES]);
int heat_batch_comm_init_ex(heat_batch *b, const uint8_t id[HEAT_COMM_ID_BYTES], const int32_t *extra_shared,
                            int32_t n_extra);
int heat_batch_set_owned_zones(heat_batch *b, const uint8_t *owned);
int32_t heat_batch_n_shared_zones(const heat_batch *b);
/* Ranks of the batch's communicator (0: it has none — single GPU, a partition that shares no zone, or a host that
 * brings its own collective). */
int32_t heat_batch_comm_ranks(const heat_batch *b);
/* Gives the communicator up (ncclCommDestroy) and returns the batch to "sharded, no communicator, shared zones not
 * agreed": for a host whose ranks found out — collectively, by their own means — that heat_batch_comm_init failed
 * on SOME rank, and that now fall back to the split-phase calls with their own collective on EVERY rank. No-op
 * without a communicator. (A failed heat_batch_comm_init[_ex] has already done this on the rank it failed on.) */
int heat_batch_comm_destroy(heat_batch *b);

/*
 * Cluster-resident march (on by default). ThermalModel::march runs its dt_subdivisions sub-timesteps back to back
 * and nothing outside reads the state in between (model.rs:369-424), and surfaces exchange heat only through the
 * zones they face (model.rs:556-590). So the batch is cut into zone-connected clusters; a cluster whose surfaces
 * are all palette-form fast-path walls (gas cavities between massive nodes and no-mass chunks of one or two nodes
 * allowed with 4 or 8 nodes per lane) or small all-no-mass surfaces, and for which the planner's cost model expects a
 * gain, is marched for ALL n_sub sub-timesteps of a heat_batch_march[_resident] call resident on the chip: node
 * temperatures stay in registers, the zone balance is summed on the chip, and only the final temperatures,
 * coefficients and flows are written. A cluster of up to eight wavefronts is one workgroup's; a larger one (a building
 * whose rooms are all joined by interior walls: up to 32 wavefronts, 256 zones) is marched by a TEAM of up to eight
 * workgroups that exchange the partial sums of the zones they share through L2 once per sub-timestep (a march that
 * could not complete that exchange returns HEAT_E_DEVICE). Everything else is streamed one sub-timestep per launch. The zone sums (here and in the streamed k_zones) are lane-strided partial sums followed
 * by a fixed reduction tree: deterministic run to run, but NOT the sequential surface order of model.rs:562-585 — the
 * results differ from a sequential sum by rounding (~1e-16 relative; everything is tested at 1e-9 against the oracle).
 * March calls of a single sub-timestep are streamed as well (the fused launch pays off from two on).
 * heat_batch_set_fusion(b, 0) streams everything (used to measure the per-sub-timestep kernel on its own). */
int heat_batch_set_fusion(heat_batch *b, int32_t enabled);
int64_t heat_batch_n_fused_surfaces(const heat_batch *b);
int64_t heat_batch_n_fused_launches(const heat_batch *b); /* cluster-resident launches issued since creation */

/* Introspection (tests, bench). */
int64_t heat_batch_n_surfaces(const heat_batch *b);
int64_t heat_batch_n_nodes(const heat_batch *b);
int64_t heat_batch_n_zones(const heat_batch *b);
/* Bytes one sub-timestep must move at minimum: 32 B per node + per-surface scalars. */
int64_t heat_batch_algorithmic_bytes(const heat_batch *b);
/* Total iterations of the no-mass fixed-point loop (surface.rs:808-896) since creation. */
int64_t heat_batch_nomass_iterations(heat_batch *b);
/* Number of surfaces routed to {fast M=4, fast M=8, fast M=16, small all-no-mass, general catch-all}. */
int heat_batch_class_counts(const heat_batch *b, int64_t counts[5]);
/* Kernel timing with HIP events recorded on the batch's stream around the surface kernels of
 * every sub-timestep executed while enabled (the march then runs eagerly, not as a graph).
 * heat_batch_get_timing synchronises and returns the mean duration in microseconds of the
 * surface kernels of one sub-timestep (*surf_us), of one whole sub-timestep (*substep_us) and
 * the number of sub-timesteps sampled; it then clears the samples.
 * enabled = k > 1: of the streamed march calls only every k-th one records events (and runs eagerly); the others
 * replay the graph as they do untimed — the rate of a timed region then stays close to the untimed one. */
int heat_batch_set_timing(heat_batch *b, int32_t enabled);
int heat_batch_get_timing(heat_batch *b, double *surf_us, double *substep_us, int64_t *n_samples);

/*
 * Partition of a model over the GPUs of a node (host-only: needs no device). Surfaces exchange heat only through
 * the zones they face (model.rs:556-590), so a shard cut ALONG the zone-connected clusters shares no zone with
 * another shard and needs no exchange at all; only a cluster heavier than a quarter of a shard (a whole building
 * whose zones are all joined by interior walls) is cut by surface ranges, sharing zones at the cuts.
 *   rank_of_surface[n_surfaces]   out: the rank every surface goes to, in [0, n_ranks); balanced by the algorithmic
 *                                 bytes of the surfaces (32 n + 152), clusters kept in model order
 *   n_shared_zones                out, nullable: zones faced by surfaces of more than one rank (0: no collective
 *                                 is ever issued by the sharded march)
 * heat_batch_create_shard builds the batch of one rank from the WHOLE model's descriptor and that partition
 * (rank = opt->rank; zones, cavities and state slots stay global).
 */
int heat_partition(const heat_batch_desc *desc, int32_t n_ranks, int32_t *rank_of_surface, int64_t *n_shared_zones);
int heat_batch_create_shard(const heat_batch_desc *desc, const heat_batch_options *opt, const int32_t *rank_of_surface,
                            heat_batch **out);
/* Host-only self-check of the planner (tests): plans `desc` as heat_batch_create_ex would and verifies the plan's
 * internal consistency (every surface in exactly one tile, every index inside its array, every workgroup of the
 * cluster-resident march inside the kernel's limits). summary (nullable): surfaces per kernel class [5], surfaces
 * in the cluster-resident march, its workgroups, tiles. */
int heat_plan_check(const heat_batch_desc *desc, const heat_batch_options *opt, int64_t summary[8]);

const char *heat_last_error(void);
int heat_amd_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
