// batch.hip — implementation of the C ABI declared in include/heat_amd.h.
//
// heat_batch_create re-orders the caller's surfaces into the lane-blocked device layout
// (layout.hpp), heat_batch_march* drive the kernels of kernels.hip on one HIP stream.
// There is no CPU fallback: every entry point that computes needs a HIP device and fails
// with HEAT_E_DEVICE otherwise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is loaded at run time (heat_batch_comm_init), never linked

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/heat_amd.h"
#include "kernels.hpp"
#include "layout.hpp"
#include "hostpool.hpp"
#include "plan.hpp"

using namespace heat;

namespace {

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    heat::last_error() = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(HEAT_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// RCCL entry points, bound at run time so that single-GPU users need no RCCL at all. In a process that
// has loaded torch, "librccl.so.1" resolves to the copy torch already mapped (same soname).
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

struct RcclState {
    Rccl r;
    std::string err;
    std::once_flag once;
};
RcclState &rccl_state() {
    static RcclState st;
    return st;
}
const std::string &rccl_error() { return rccl_state().err; }

Rccl *rccl() {
    RcclState &st = rccl_state();
    std::call_once(st.once, [&st] {
        Rccl &r = st.r;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
            const char *e = dlerror();  // (read once: dlerror() clears the message)
            st.err = e ? e : "dlopen failed";
        }
        if (!h) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce || !r.GetErrorString) {
            st.err = "symbols missing";
            dlclose(h);
            return;
        }
        st.err.clear();
        r.handle = h;
    });
    return st.r.handle ? &st.r : nullptr;
}

#define RCCL_TRY(R, expr)                                                                             \
    do {                                                                                              \
        ncclResult_t e_ = (expr);                                                                     \
        if (e_ != ncclSuccess)                                                                        \
            return fail(HEAT_E_COMM, "%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
    }
    hipError_t upload(const std::vector<T> &h) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    hipError_t zeros(size_t count) {
        hipError_t e = alloc(count);
        if (e != hipSuccess || count == 0) return e;
        // (the batch's streams do not wait for the null stream — hipStreamNonBlocking —: the fill must be over before
        // any of them may touch the buffer. Found by tools/fuzz.py: the teams' exchange areas, allocated with the first team
        // launch, were still being zeroed while the members published into them.)
        e = hipMemset(p, 0, count * sizeof(T));
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(nullptr);
    }
};

// march calls of fewer sub-timesteps are streamed (large batches). Round 3, 1 M x 32 on one MI355X (tools/short_calls.py): a
// cluster-resident call costs 155 us + 59 us per sub-timestep, the same batch streamed 198-212 us per sub-timestep (its layout
// pads every cluster to whole wavefronts; laid out for streaming alone it takes 166-188) -> resident from two sub-timesteps on.
static const int kFusedMinSubsteps = getenv("HEAT_AMD_FUSED_MIN") ? atoi(getenv("HEAT_AMD_FUSED_MIN")) : 2;

}  // namespace

struct heat_batch {
    int device = 0;
    int n_cu = 256;  // compute units of `device` (sizes the persistent grids and the fused launches' room)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // Independent surface classes run on side streams between a fork and a join event (captured into
    // the sub-timestep graph as parallel branches).
    static constexpr int kSideStreams = 3;
    hipStream_t side[kSideStreams] = {nullptr, nullptr, nullptr};
    hipStream_t fused_stream = nullptr;  // the cluster-resident march when other surfaces are streamed beside it
    hipEvent_t ev_fork = nullptr, ev_join[kSideStreams] = {nullptr, nullptr, nullptr};
    int n_ranks = 1, rank = 0;
    bool use_graph = false;

    int64_t n_surf = 0, n_zones = 0, n_state = 0, n_nodes = 0, n_cav = 0;
    double dt = 0;
    int64_t algorithmic_bytes = 0;
    int64_t class_counts[5] = {0, 0, 0, 0, 0};  // M4, M8, M16, small, general
    int64_t n_palette = 0;                      // surfaces whose constants are in palette form

    // layout
    int64_t n_fused_surfaces = 0;
    int n_stream_tiles[kNumFast] = {};  // tiles marched one sub-timestep per launch; the fused workgroups' tiles follow
    // cluster-resident march: workgroups per class, in two width groups (<= 4 tiles, <= 8 tiles)
    // workgroup lists per class: index = width group (0: <= 4 wavefronts, 1: <= 8) + 2 * mixed (small-surface tiles too)
    std::vector<FusedBlock> h_fblocks[kNumFast][4];
    DevBuf<FusedBlock> d_fblocks[kNumFast][4];
    // teams of workgroups (clusters larger than one workgroup; layout.hpp, FusedSuper)
    std::vector<FusedBlock> h_team_blocks[kNumFast], h_team_blocks0[kNumFast];
    std::vector<FusedSuper> h_team_supers[kNumFast];
    DevBuf<FusedBlock> d_team_blocks[kNumFast];
    DevBuf<FusedSuper> d_team_supers[kNumFast];
    DevBuf<uint32_t> d_team_zinfo;
    DevBuf<unsigned long long> d_xbuf;  // the teams' exchange areas (allocated with the first team launch)
    int xbuf_teams = 0;                 // teams the exchange areas are sized for
    uint32_t team_epoch = 0;            // launches so far: names the granules of a launch (tag_base)
    bool any_teams = false;
    DevBuf<unsigned int> d_fqueue;      // work-queue counters of the fused launches (sharded batches): one per list
    DevBuf<int32_t> d_fzones, d_fzone_eoff;
    DevBuf<uint16_t> d_fslots;
    DevBuf<double> d_side_area;
    DevBuf<int16_t> d_side_lzone;
    // the plan as made at create time (heat_batch_set_shared_zones derives the current one from it: a workgroup
    // that owns a zone shared with another rank is demoted — its tiles and zones are streamed, so that the zone
    // exchange can happen every sub-timestep)
    std::vector<FastTile> h_tiles0[kNumFast];
    std::vector<FusedBlock> h_fblocks0[kNumFast][4];
    int n_stream_tiles0[kNumFast] = {};
    std::vector<int32_t> h_fzones;      // global zone of fused-zone index i (FusedBlock::first_zone + j)
    DevBuf<int32_t> d_zlist_stream;     // sharded: touched zones that no fused workgroup owns (shared ones first)
    int n_touched_stream = 0;
    std::vector<int32_t> h_zone_block;  // zone -> fused workgroup (global number) or -1
    DevBuf<int32_t> d_stream_zones;     // zones whose balance is done by k_zones (not owned by a fused workgroup)
    int n_stream_zones = 0;
    bool any_fused = false;
    hipEvent_t ev_fused = nullptr;
    hipEvent_t ev_staged = nullptr;  // the H2D copies of a march call's weather / zone terms have run
    bool staged = false;
    int n_fast_tiles[kNumFast] = {};
    DevBuf<FastTile> d_fast_tiles[kNumFast];
    std::vector<FastTile> h_tiles_cur[kNumFast];  // the tile lists as they are on the device now
    // Unified streamed lists (k_surfaces_stream): the palette-form fast classes and the cavity-free small surfaces in
    // one tile list. [0]: the streamed tiles only (beside a cluster-resident march), [1]: every tile.
    DevBuf<FastTile> d_ulist[2];
    int n_ulist[2] = {0, 0};
    // ... in three parts, one per variant of the kernel (kernels.hip: 16 nodes per lane | 8 / 4 nodes per lane and small
    // surfaces | tiles with no-mass chunks other than facings), launched back to back: [part] = first tile, tiles
    int ulist_part[2][kStreamVariants][2] = {};
    bool capturing = false;         // a stream capture of the batch's stream is under way (enqueue_surfaces does not fork then)
    bool host_zones_stale = false;  // the device has marched since the caller's state last received the zone temperatures
    unsigned int sweep_parity = 0;  // enqueue_surfaces: direction of the next streamed sweep (zig-zag)
    unsigned int fused_parity = 0;  // enqueue_fused: direction of the next cluster-resident launch
    bool class_has_chunks[kNumFast] = {};  // the class holds tiles with such chunks: its own launch takes the NM = 2 variant
    bool in_ulist[2][kNumFast] = {};
    bool small_in_ulist[2] = {false, false};
    bool smallcav_in_ulist[2] = {false, false};  // the double glazing (small surfaces with a gas cavity) as well
    DevBuf<unsigned long long> d_ucount;  // no-mass pass counters of the unified lists: [list][tile]
    size_t ucount_stride = 0;
    int n_gen_tiles = 0;    // tiles in the general layout: [0, n_small_tiles) small, the rest catch-all
    int n_small_tiles = 0;      // small tiles, cavity-free ones first
    int n_small_plain_tiles = 0;
    int n_smallcav_stream_tiles = 0;  // streamed small-with-cavity tiles; the fused workgroups' small tiles follow them
    int n_smallcav_stream_tiles0 = 0;
    std::vector<GeneralTile> h_gen_tiles0;
    std::vector<GeneralTile> h_gen_tiles_cur;  // as on the device now
    size_t nm_count_base[kNumFast + 1] = {};
    DevBuf<GeneralTile> d_gen_tiles;
    int64_t gen_base = 0;     // first node slot of the general group
    int64_t node_slots = 0;   // total node slots incl. padding

    DevBuf<double> d_T, d_V, d_U, d_alpha_f, d_alpha_b, d_mass, d_scratch;
    DevBuf<int32_t> d_cav_idx;
    DevBuf<int32_t> d_cavref;  // CAV fast classes: 4 ints per device surface
    DevBuf<uint8_t> d_cls;   // palette class bytes (PAL fast classes)
    DevBuf<double> d_pal;    // palettes, na.pal_stride doubles per device surface
    DevBuf<CavityDev> d_cavs;

    DevBuf<int32_t> d_meta;        // node count per device surface (upload/download kernels)
    DevBuf<SideConst> d_side_const;  // [2 * S]: front records, then back records
    DevBuf<double> d_side_alpha;     // [2 * S]: factor applied to the solar irradiance at upload (layout.hpp, SideDyn)
    DevBuf<SideDyn> d_side_dyn;      // [2 * S]
    DevBuf<SideOut> d_side_out;      // [2 * S]
    DevBuf<double> d_hs_fix;         // [2 * S] or empty
    DevBuf<int64_t> d_first_slot, d_slots;  // d_slots: 8 arrays of n_surf
    DevBuf<int64_t> d_zone_slot, d_zone_off;
    DevBuf<ZoneEntry> d_zone_entries;
    int zone_rows = 0;  // k_zones gives a zone a row of 16 lanes (few walls per zone) instead of a wavefront
    DevBuf<ZoneContrib> d_zone_contrib;  // [zone entries], written by the surface kernels
    DevBuf<double> d_zone_vol, d_zone_T, d_zone_a0, d_zone_b0, d_partial;
    double *partial_ptr = nullptr;  // where step_surfaces writes (a, b): d_partial or caller memory
    // sharded batches (heat_batch_set_shared_zones)
    bool shared_set = false;
    int n_shared = 0, n_touched = 0;
    DevBuf<int32_t> d_zlist, d_slot_of, d_shared_zone;
    std::vector<int64_t> h_orig_of;  // device surface -> surface of the caller's descriptor
    int64_t fail_index = -1;         // where the last reported numerical failure happened first (heat_batch_failed_surface)
    int32_t fail_kind = 0;
    std::vector<uint8_t> h_touched;
    // Zones this batch finishes itself (unless they are shared with another rank): every zone on a single GPU; on a
    // sharded batch the zones its surfaces face plus the zones NO rank faces that fall to it (z % n_ranks == rank) —
    // such a zone still follows its a0 / b0 terms (heaters, infiltration: model.rs:410-423).
    std::vector<uint8_t> h_owned;
    // library-owned collective (heat_batch_comm_init): RCCL communicator + the gathered partial blocks
    ncclComm_t comm = nullptr;
    DevBuf<double> d_gathered;  // [n_ranks][2][n_shared]
    DevBuf<double> d_state;
    DevBuf<StepWeather> d_weather;
    DevBuf<int> d_step, d_flags;
    int *h_flags = nullptr;  // pinned: the flags as of the last heat_batch_synchronize
    DevBuf<unsigned long long> d_nomass_iters;

    StepWeather *h_weather = nullptr;  // pinned
    double *h_zone_ab = nullptr;       // pinned, [2][n_zones]
    size_t weather_cap = 0;
    int n_weather = 0;

    // heat_batch_march on a caller-owned state: compact transfers through pinned staging, host gathers / scatters
    // on a thread pool ("Data at the boundary")
    std::vector<int64_t> h_in_slots;   // [4][S] in device surface order: solar_f, solar_b, ir_f, ir_b
    std::vector<int64_t> h_node_off;   // [S + 1] node offsets in the caller's surface order
    DevBuf<double> d_compact;          // inputs [4 S + Z]; outputs [N nodes | 4 S scalars | Z zones]
    DevBuf<int64_t> d_compact_off;     // per device surface: where its nodes go in the compact output
    DevBuf<int32_t> d_orig_of32;
    double *h_pin = nullptr;           // pinned staging
    size_t pin_doubles = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy[2] = {nullptr, nullptr};
    std::vector<std::pair<int64_t, int64_t>> out_chunks;  // surface ranges of the node part, a staging half each
    HostPool *pool = nullptr;
    // Runs of the state in which EVERY slot is an output of this path or an input it was just handed (the reference's
    // layout: one run over all surface blocks, surface_trait.rs:223-378). heat_batch_march copies them from the state
    // mirror straight into the caller's array — no staging, no host scatter. Empty: the layout is not of that kind.
    std::vector<std::pair<int64_t, int64_t>> direct_runs;

    // host copies for download (original surface order)
    std::vector<int64_t> h_first_slot, h_node_count, h_out_slots[4], h_zone_slot_h;
    std::vector<double> h_stage;

    SideArrays sa{};
    NodeArrays na{};
    SlotArrays sl{};

    // graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;

    // timing
    bool timing = false;
    int timing_every = 1;      // heat_batch_set_timing(k > 1): only every k-th streamed march call records events
    int64_t timing_calls = 0;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct EvTriple { hipEvent_t e0, e1, e2; };            // one streamed sub-timestep: start, surfaces done, zones done
    struct EvPair { hipEvent_t e0, e1; int n_sub; };       // one cluster-resident launch over n_sub sub-timesteps
    std::vector<EvTriple> ev_triples;
    std::vector<EvPair> ev_fused_pairs;
    bool fusion_on = true;     // heat_batch_options::no_fusion / heat_batch_set_fusion
    int64_t n_fused_launches = 0;  // cluster-resident launches issued since creation (introspection)
    bool graph_fused = false;  // what the captured sub-timestep graph leaves out
    int graph_subs = 0;        // sub-timesteps one replay of the captured graph runs
    int graph_recaptures = 0;  // captures forced by a change of the calls' length

    ~heat_batch() {
        if (comm) {
            Rccl *r = rccl();
            if (r) (void)r->CommDestroy(comm);
        }
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph) (void)hipGraphDestroy(graph);
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        delete pool;
        if (h_pin) (void)hipHostFree(h_pin);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (auto e : ev_copy) if (e) (void)hipEventDestroy(e);
        if (h_weather) (void)hipHostFree(h_weather);
        if (h_zone_ab) (void)hipHostFree(h_zone_ab);
        if (h_flags) (void)hipHostFree(h_flags);
        for (int i = 0; i < kSideStreams; i++) {
            if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
            if (side[i]) (void)hipStreamDestroy(side[i]);
        }
        if (fused_stream) (void)hipStreamDestroy(fused_stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_fused) (void)hipEventDestroy(ev_fused);
        if (ev_staged) (void)hipEventDestroy(ev_staged);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

int flags_to_status(int f) {
    if (f & FLAG_NAN_HS) return fail(HEAT_N_NAN_HS, "NaN convection coefficient (reference: surface.rs:704)");
    if (f & FLAG_NAN_NOMASS) return fail(HEAT_N_NAN_NOMASS, "NaN error in the no-mass loop (reference: surface.rs:850)");
    if (f & FLAG_NAN_ZONE) return fail(HEAT_N_NAN_ZONE, "NaN zone temperature (reference: model.rs:417)");
    if (f & FLAG_UNREACHABLE) return fail(HEAT_N_UNREACHABLE, "unreachable!() branch taken (NaN input)");
    if (f & FLAG_EXCHANGE)
        return fail(HEAT_E_DEVICE, "cluster-resident march: a workgroup of a team gave up waiting for another member's zone sums "
                                   "(the state of this march call is not to be used)");
    return HEAT_OK;
}

int rebuild_unified(heat_batch *b);

int build(heat_batch *b, const heat_batch_desc *d, const heat_batch_options &opt) {
    // Classification, clustering, tiling and packing are host-only work (plan.cpp); here the plan is uploaded.
    Plan p;
    {
        std::string err;
        const int rc = make_plan(d, opt, p, err);
        if (rc) return fail(rc, "%s", err.c_str());
    }
    const int64_t S = p.n_surf, Z = p.n_zones;
    b->n_surf = S;
    b->n_zones = Z;
    b->n_state = p.n_state;
    b->n_cav = p.n_cav;
    b->dt = p.dt;
    b->n_nodes = p.n_nodes;
    b->algorithmic_bytes = p.algorithmic_bytes;
    for (int i = 0; i < 5; i++) b->class_counts[i] = p.class_counts[i];
    b->n_palette = p.n_palette;
    b->n_fused_surfaces = p.n_fused_surfaces;
    b->gen_base = p.gen_base;
    b->node_slots = p.node_slots;
    b->n_small_tiles = p.n_small_tiles;
    b->n_small_plain_tiles = p.n_small_plain_tiles;
    b->n_smallcav_stream_tiles = b->n_smallcav_stream_tiles0 = p.n_smallcav_stream_tiles;
    b->any_fused = p.any_fused;
    b->h_zone_block = p.zone_block;
    b->h_gen_tiles0 = b->h_gen_tiles_cur = p.gen_tiles;
    for (int c = 0; c < kNumFast; c++) {
        for (int g2 = 0; g2 < 4; g2++) {
            b->h_fblocks[c][g2] = b->h_fblocks0[c][g2] = p.fblocks[c][g2];
            HIP_TRY(b->d_fblocks[c][g2].upload(p.fblocks[c][g2]));
        }
        b->h_team_blocks[c] = b->h_team_blocks0[c] = p.team_blocks[c];
        b->h_team_supers[c] = p.team_supers[c];
        b->any_teams = b->any_teams || !p.team_supers[c].empty();
        HIP_TRY(b->d_team_blocks[c].upload(p.team_blocks[c]));
        HIP_TRY(b->d_team_supers[c].upload(p.team_supers[c]));
        b->h_tiles0[c] = b->h_tiles_cur[c] = p.fast_tiles[c];
        b->n_stream_tiles[c] = b->n_stream_tiles0[c] = p.n_stream_tiles[c];
        b->n_fast_tiles[c] = (int)p.fast_tiles[c].size();
        HIP_TRY(b->d_fast_tiles[c].upload(p.fast_tiles[c]));
        b->nm_count_base[c] = p.nm_count_base[c];
    }
    b->nm_count_base[kNumFast] = p.nm_count_base[kNumFast];
    b->h_fzones = p.fzones;
    HIP_TRY(b->d_fzones.upload(p.fzones));
    HIP_TRY(b->d_fzone_eoff.upload(p.fzone_eoff));
    HIP_TRY(b->d_team_zinfo.upload(p.team_zinfo));
    HIP_TRY(b->d_fslots.upload(p.fslots));
    HIP_TRY(b->d_side_area.upload(p.side_area));
    HIP_TRY(b->d_side_lzone.upload(p.side_lzone));
    b->h_touched = p.touched;
    b->h_owned = p.touched;
    if (b->n_ranks <= 1) b->h_owned.assign(Z, 1);
    {
        std::vector<int32_t> sz;
        for (int64_t z = 0; z < Z; z++) if (b->h_zone_block[z] < 0 && b->h_owned[z]) sz.push_back((int32_t)z);
        b->n_stream_zones = (int)sz.size();
        HIP_TRY(b->d_stream_zones.upload(sz));
    }

    // ---- host copies used by download ----
    b->h_first_slot = std::move(p.h_first_slot);
    b->h_node_count = std::move(p.h_node_count);
    for (int a = 0; a < 4; a++) b->h_out_slots[a] = std::move(p.h_out_slots[a]);
    b->h_zone_slot_h = p.zone_slot;

    // ---- upload ----
    b->n_gen_tiles = (int)p.gen_tiles.size();
    HIP_TRY(b->d_gen_tiles.upload(p.gen_tiles));
    HIP_TRY(b->d_T.zeros(p.node_slots));
    HIP_TRY(b->d_V.upload(p.V));
    HIP_TRY(b->d_U.upload(p.U));
    HIP_TRY(b->d_alpha_f.upload(p.alpha_f));
    HIP_TRY(b->d_alpha_b.upload(p.alpha_b));
    HIP_TRY(b->d_mass.upload(p.mass));
    HIP_TRY(b->d_cav_idx.upload(p.cav_idx));
    HIP_TRY(b->d_cavref.upload(p.cavref));
    HIP_TRY(b->d_cls.upload(p.cls));
    HIP_TRY(b->d_pal.upload(p.pal));
    HIP_TRY(b->d_scratch.alloc(p.scratch_slots));
    HIP_TRY(b->d_cavs.upload(p.cavs));
    HIP_TRY(b->d_meta.upload(p.meta));
    HIP_TRY(b->d_side_const.upload(p.side));
    HIP_TRY(b->d_side_alpha.upload(p.side_alpha));
    const bool has_fix = !p.hs_fix.empty();
    if (has_fix) HIP_TRY(b->d_hs_fix.upload(p.hs_fix));
    HIP_TRY(b->d_side_dyn.zeros(2 * S));
    HIP_TRY(b->d_side_out.zeros(2 * S));
    HIP_TRY(b->d_first_slot.upload(p.first_slot));
    HIP_TRY(b->d_slots.upload(p.slots));
    HIP_TRY(b->d_zone_slot.upload(p.zone_slot));
    HIP_TRY(b->d_zone_vol.upload(p.zone_vol));
    HIP_TRY(b->d_zone_off.upload(p.zone_off));
    HIP_TRY(b->d_zone_entries.upload(p.zone_entries));
    {
        static const char *rows_env = getenv("HEAT_AMD_ZONE_ROWS");  // measurement: 0 / 1 overrides the rule
        b->zone_rows = rows_env ? atoi(rows_env) : ((int64_t)p.zone_entries.size() <= 24 * (int64_t)b->n_zones);
    }
    HIP_TRY(b->d_zone_contrib.zeros(p.zone_entries.size()));
    HIP_TRY(b->d_zone_T.zeros(std::max<int64_t>(Z, 1)));  // (never empty: the surface kernels gather unconditionally)
    HIP_TRY(b->d_zone_a0.zeros(Z));
    HIP_TRY(b->d_zone_b0.zeros(Z));
    HIP_TRY(b->d_partial.zeros(2 * Z));
    b->partial_ptr = b->d_partial.p;
    HIP_TRY(b->d_state.zeros(p.n_state));
    HIP_TRY(b->d_step.zeros(2));  // [0] sub-timestep of the running march call, [1] its last one
    HIP_TRY(b->d_flags.zeros(4));  // [0] kinds OR-ed, [2..3] first failing surface / zone (report_failure, kernels.hip)
    HIP_TRY(hipMemset(b->d_flags.p + 2, 0xff, 2 * sizeof(int)));
    HIP_TRY(hipStreamSynchronize(nullptr));
    b->h_orig_of = p.orig_of;
    HIP_TRY(b->d_nomass_iters.zeros(p.n_nm_counters));
    if (Z > 0) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_zone_ab), 2 * Z * sizeof(double)));

    {
        const int rc = rebuild_unified(b);
        if (rc) return rc;
    }
    {   // compact transfers
        b->h_in_slots.assign(p.slots.begin() + 4 * S, p.slots.end());
        b->h_node_off.assign(d->node_offset, d->node_offset + S + 1);
        if (S == 0) b->h_node_off.assign(1, 0);
        std::vector<int64_t> coff(S);
        std::vector<int32_t> oo(S);
        for (int64_t dd = 0; dd < S; dd++) {
            coff[dd] = d->node_offset[p.orig_of[dd]];
            oo[dd] = (int32_t)p.orig_of[dd];
        }
        HIP_TRY(b->d_compact_off.upload(coff));
        HIP_TRY(b->d_orig_of32.upload(oo));
        const size_t n_out = (size_t)p.n_nodes + 4 * (size_t)S + (size_t)Z;
        HIP_TRY(b->d_compact.alloc(std::max<size_t>(n_out, 1)));
        // staging: two halves of up to 4 M doubles (32 MB) — a copy fills one while the pool empties the other; the
        // inputs (4 S + Z) and the scalar outputs (4 S + Z) must fit one buffer as a whole
        static const int stage_mb = getenv("HEAT_AMD_STAGE_MB") ? atoi(getenv("HEAT_AMD_STAGE_MB")) : 32;  // per half
        const size_t kHalf = (size_t)std::max(stage_mb, 1) << 17;
        b->pin_doubles = std::max<size_t>(2 * std::min<size_t>(kHalf, std::max<size_t>((size_t)p.n_nodes, 1)), 4 * (size_t)S + (size_t)Z + 1);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_pin), b->pin_doubles * sizeof(double)));
        const size_t half = b->pin_doubles / 2;
        b->out_chunks.clear();
        for (int64_t s0 = 0; s0 < S;) {
            int64_t s1 = s0 + 1;
            while (s1 < S && (size_t)(d->node_offset[s1 + 1] - d->node_offset[s0]) <= half) s1++;
            b->out_chunks.push_back({s0, s1});
            s0 = s1;
        }
        HIP_TRY(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
        for (auto &e : b->ev_copy) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        {   // direct runs (see heat_batch::direct_runs)
            static const bool no_direct = getenv("HEAT_AMD_NO_DIRECT_DOWNLOAD") != nullptr;
            std::vector<uint8_t> mask((size_t)p.n_state, 0);
            const int64_t *sc[8] = {d->hs_front_slot, d->hs_back_slot, d->flow_front_slot, d->flow_back_slot,
                                    d->solar_front_slot, d->solar_back_slot, d->ir_front_slot, d->ir_back_slot};
            for (int64_t s = 0; s < S; s++) {
                const int64_t n = d->node_offset[s + 1] - d->node_offset[s];
                memset(&mask[d->first_node_slot[s]], 1, (size_t)n);
                for (int a = 0; a < 8; a++) mask[sc[a][s]] = 1;
            }
            int64_t covered = 0, wanted = 0;
            for (int64_t i = 0; i < p.n_state;) {
                if (!mask[i]) { i++; continue; }
                int64_t e = i;
                while (e < p.n_state && mask[e]) e++;
                wanted += e - i;
                if (e - i >= 65536) { b->direct_runs.push_back({i, e}); covered += e - i; }
                i = e;
            }
            // all of the surfaces' slots in a few long runs, or nothing (the staged path serves any layout)
            if (no_direct || covered != wanted || b->direct_runs.size() > 64) b->direct_runs.clear();
        }
        static const int n_env = getenv("HEAT_AMD_HOST_THREADS") ? atoi(getenv("HEAT_AMD_HOST_THREADS")) : 0;
        const int hw = (int)std::thread::hardware_concurrency();
        b->pool = new HostPool(n_env > 0 ? n_env : std::max(1, std::min(hw > 0 ? hw : 8, S > 20000 ? 16 : 1)));
        if (getenv("HEAT_AMD_TRACE")) fprintf(stderr, "heat_amd: %d host threads, staging 2 x %zu MB, %zu node pieces\n", b->pool->size(), half * 8 >> 20, b->out_chunks.size());
    }

    // ---- argument bundles ----
    SideArrays &sa = b->sa;
    sa.sc = b->d_side_const.p;
    sa.dyn = b->d_side_dyn.p;
    sa.out = b->d_side_out.p;
    sa.hs_fix = has_fix ? b->d_hs_fix.p : nullptr;
    sa.zc = b->d_zone_contrib.p;
    sa.S = (int32_t)S;
    sa.pad = 0;
    NodeArrays &na = b->na;
    na.T = b->d_T.p; na.V = b->d_V.p; na.U = b->d_U.p;
    na.cls = b->d_cls.p; na.pal = b->d_pal.p;
    na.pal_stride = p.pal_stride;
    na.pal_ubase = p.pal_ubase;
    na.cavref = b->d_cavref.p; na.cavs = b->d_cavs.p;
    na.alpha_f = b->d_alpha_f.p; na.alpha_b = b->d_alpha_b.p; na.cav = b->d_cav_idx.p; na.mass = b->d_mass.p;
    SlotArrays &sl = b->sl;
    const int64_t *sp = b->d_slots.p;
    sl.hs_f = sp; sl.hs_b = sp + S; sl.flow_f = sp + 2 * S; sl.flow_b = sp + 3 * S;
    sl.solar_f = sp + 4 * S; sl.solar_b = sp + 5 * S; sl.ir_f = sp + 6 * S; sl.ir_b = sp + 7 * S;
    // (every upload and fill above went through the null stream; the batch's streams do not wait for it)
    HIP_TRY(hipDeviceSynchronize());
    return HEAT_OK;
}

int select_device(heat_batch *b) {
    HIP_TRY(hipSetDevice(b->device));
    return HEAT_OK;
}

// A persistent wavefront w of a k_surfaces_stream launch walks tiles[w], tiles[w + n_waves], ... — round after round. The
// tiles of a part differ in length by an order of magnitude (a window whose no-mass loop re-evaluates its gas cavity
// every pass against eight massive nodes per lane), so the list is laid out in rounds such that every wavefront's share
// takes about the same time: longest-processing-time-first over the wavefronts, each wavefront's tiles longest first,
// empty tiles (G = 0: a wavefront skips them at once) where a wavefront has fewer tiles than rounds.
std::vector<FastTile> balanced_rounds(const std::vector<FastTile> &tiles, const std::vector<float> &cost, int n_waves) {
    const size_t n = tiles.size();
    if (n_waves <= 0 || n <= (size_t)n_waves) return tiles;  // a tile per wavefront at most: nothing to balance
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cost[a] > cost[b]; });
    // least-loaded wavefront first (a binary heap of (load, wavefront))
    std::vector<std::pair<float, int32_t>> heap(n_waves);
    for (int w = 0; w < n_waves; w++) heap[w] = {0.f, w};
    auto later = [](const std::pair<float, int32_t> &a, const std::pair<float, int32_t> &b) {
        return a.first > b.first || (a.first == b.first && a.second > b.second);
    };
    std::make_heap(heap.begin(), heap.end(), later);
    std::vector<std::vector<int32_t>> mine(n_waves);
    for (int32_t t : order) {
        std::pop_heap(heap.begin(), heap.end(), later);
        auto &top = heap.back();
        mine[top.second].push_back(t);
        top.first += cost[t];
        std::push_heap(heap.begin(), heap.end(), later);
    }
    size_t rounds = 0;
    for (const auto &m : mine) rounds = std::max(rounds, m.size());
    FastTile empty;
    empty.node_base = 0;
    empty.surf_base = 0;
    empty.k = (int16_t)(kStreamKindSmall << kStreamKindShift);  // (a small tile of no surface: every lane leaves at once)
    empty.G = 0;
    std::vector<FastTile> out(rounds * (size_t)n_waves, empty);
    for (int w = 0; w < n_waves; w++)
        for (size_t r = 0; r < mine[w].size(); r++) out[r * (size_t)n_waves + w] = tiles[mine[w][r]];
    return out;
}

// The unified streamed tile lists from the current per-class lists (create time, and again when
// heat_batch_set_shared_zones has moved tiles between the fused workgroups and the streamed part).
int rebuild_unified(heat_batch *b) {
    static const bool off = getenv("HEAT_AMD_NO_UNIFIED") != nullptr;  // measurement: one launch per class as before
    // (16 nodes per lane with facings or cavities keep launches of their own: one wavefront per SIMD)
    auto eligible = [](int c) { return kFastPAL[c] && !(kFastM[c] == 16 && (kFastNM[c] || kFastCAV[c])); };
    static const bool cav_off = getenv("HEAT_AMD_NO_UNIFIED_CAV") != nullptr;  // measurement: round 2's two launches
    size_t cap = (size_t)b->n_small_tiles;
    for (int c = 0; c < kNumFast; c++) if (eligible(c)) cap += b->h_tiles_cur[c].size();
    // (the balanced parts are padded to whole rounds: two rounds more than the tiles need, per part, bound the padding)
    const int waves_light = b->n_cu * 3 * 4, waves_cav = b->n_cu * 2 * 4;  // persistent grids of the light / cavity variants
    cap += 2 * (size_t)(waves_light + waves_cav) + 2 * (size_t)(waves_light + waves_cav);
    if (b->d_ucount.n == 0) {
        b->ucount_stride = std::max<size_t>(cap, 1);
        HIP_TRY(b->d_ucount.zeros(2 * b->ucount_stride));
    }
    const int ns = b->n_small_plain_tiles;
    for (int c = 0; c < kNumFast; c++) {
        b->class_has_chunks[c] = false;
        for (const FastTile &ft : b->h_tiles_cur[c]) b->class_has_chunks[c] = b->class_has_chunks[c] || (ft.k & kTileChunkyBit) != 0;
    }
    for (int v = 0; v < 2; v++) {
        // the four parts (kernels.hip, k_surfaces_stream): wide | light | chunks | cavities
        std::vector<FastTile> part[kStreamVariants];
        int n_classes = 0, n_cav_classes = 0;
        // double glazing: the streamed small tiles with cavities ([plain | streamed with cavities | fused workgroups' ...])
        const int nsc = cav_off ? 0 : (v ? b->n_small_tiles - b->n_small_plain_tiles : b->n_smallcav_stream_tiles);
        for (int j = 0; j < nsc; j++) {  // (first in their part: the longest tiles)
            const GeneralTile &g = b->h_gen_tiles_cur[b->n_small_plain_tiles + j];
            FastTile ft;
            ft.node_base = g.node_base;
            ft.surf_base = g.surf_base;
            ft.k = (int16_t)(kStreamKindSmall << kStreamKindShift);
            ft.G = (int16_t)g.G;
            part[3].push_back(ft);
        }
        for (int mi = 2; mi >= 0; mi--)  // 16 nodes per lane first: the heaviest tiles lead
            for (int c = mi * 6; c < mi * 6 + 6; c++) {
                if (!eligible(c) || (kFastCAV[c] && cav_off)) continue;
                if (kFastCAV[c]) {
                    const int n = v ? b->n_fast_tiles[c] : b->n_stream_tiles[c];
                    n_cav_classes += n > 0;
                    for (int t = 0; t < n; t++) {
                        FastTile ft = b->h_tiles_cur[c][t];
                        ft.k = (int16_t)((ft.k & (0x1ff | kTileMixedBit | kTileChunkyBit)) | (mi << kStreamKindShift) | (kFastNM[c] ? kStreamNmBit : 0));
                        part[3].push_back(ft);
                    }
                    continue;
                }
                const int n = v ? b->n_fast_tiles[c] : b->n_stream_tiles[c];
                n_classes += n > 0;
                for (int t = 0; t < n; t++) {
                    FastTile ft = b->h_tiles_cur[c][t];
                    const bool chunks = (ft.k & kTileChunkyBit) != 0;
                    ft.k = (int16_t)((ft.k & (0x1ff | kTileMixedBit | kTileChunkyBit)) | (mi << kStreamKindShift) | (kFastNM[c] ? kStreamNmBit : 0));
                    part[mi == 2 ? 0 : (chunks ? 2 : 1)].push_back(ft);
                }
            }
        // worth it when it replaces two launches or more (a class on its own keeps its own persistent kernel)
        const bool use = !off && (n_classes + (ns > 0)) >= 2;
        // ... and the cavity part when it replaces two launches (glazing + walls with cavities, or two such classes)
        const bool use_cav = !off && !cav_off && (n_cav_classes + (nsc > 0)) >= 2;
        b->n_ulist[v] = 0;
        for (int c = 0; c < kNumFast; c++) b->in_ulist[v][c] = eligible(c) && (kFastCAV[c] ? use_cav : use);
        b->small_in_ulist[v] = use && ns > 0;
        b->smallcav_in_ulist[v] = use_cav && nsc > 0;
        for (int q = 0; q < kStreamVariants; q++) b->ulist_part[v][q][0] = b->ulist_part[v][q][1] = 0;
        if (!use) part[0].clear(), part[1].clear(), part[2].clear();
        if (!use_cav) part[3].clear();
        if (!use && !use_cav) continue;
        // Relative tile times (measured on MI355X, 1 M ragged walls / 100 k windows + 100 k Trombe walls; only their
        // ratios matter): a tile of eight massive nodes per lane 1, with facings 2, four nodes per lane 0.7 / 1.4, a tile
        // of small surfaces 3, of double glazing 12, a tile of walls with cavities 2.5.
        static const bool balance_off = getenv("HEAT_AMD_NO_BALANCE") != nullptr;  // measurement
        auto tile_cost = [&](const FastTile &ft, bool cav) -> float {
            const int kind = (ft.k >> kStreamKindShift) & 3;
            if (kind == kStreamKindSmall) return cav ? 12.f : 3.f;
            const float base = kind == 1 ? 1.f : 0.7f;
            if (cav) return 2.5f * base;
            return (ft.k & kStreamNmBit) ? 2.f * base : base;
        };
        if (use) {  // the small surfaces join the light part
            for (int j = 0; j < ns; j++) {
                const GeneralTile &g = b->h_gen_tiles_cur[j];
                FastTile ft;
                ft.node_base = g.node_base;
                ft.surf_base = g.surf_base;
                ft.k = (int16_t)(kStreamKindSmall << kStreamKindShift);
                ft.G = (int16_t)g.G;
                part[1].push_back(ft);
            }
        }
        for (int q : {1, 3}) {
            if (balance_off || part[q].empty()) continue;
            std::vector<float> cost(part[q].size());
            for (size_t t = 0; t < part[q].size(); t++) cost[t] = tile_cost(part[q][t], q == 3);
            part[q] = balanced_rounds(part[q], cost, q == 1 ? waves_light : waves_cav);
        }
        std::vector<FastTile> all;
        for (int q = 0; q < kStreamVariants; q++) {
            b->ulist_part[v][q][0] = (int)all.size();
            b->ulist_part[v][q][1] = (int)part[q].size();
            all.insert(all.end(), part[q].begin(), part[q].end());
        }
        if (all.size() > b->ucount_stride) return fail(HEAT_E_SIZE, "unified tile list grew beyond its counters");
        b->n_ulist[v] = (int)all.size();
        HIP_TRY(b->d_ulist[v].upload(all));
    }
    return HEAT_OK;
}

// iterate_surfaces for every group (model.rs:388-408), one sub-timestep.
// streamed_only: leave out the tiles owned by the cluster-resident march (enqueue_fused marches those).
void enqueue_surfaces(heat_batch *b, int step_fixed, bool streamed_only = false) {
    int nt[kNumFast];
    for (int c = 0; c < kNumFast; c++) nt[c] = streamed_only ? b->n_stream_tiles[c] : b->n_fast_tiles[c];
    // One launch for the palette-form fast classes and the cavity-free small surfaces (k_surfaces_stream); the
    // classes it does not hold follow on their own.
    const int ul = streamed_only ? 0 : 1;
    const bool unified = b->n_ulist[ul] > 0;
    // Zig-zag sweeps: consecutive sub-timesteps walk the batch in opposite directions. A sweep streams the whole state
    // through the 256 MB memory-side cache (MI355X: Infinity Cache in front of HBM); walked in the same direction every
    // time, nothing of it is left when its turn comes again — walked back, the last quarter of a 1 GB sweep is.
    static const bool zigzag_off = getenv("HEAT_AMD_NO_ZIGZAG") != nullptr;  // measurement
    const int rev = (!zigzag_off && (b->sweep_parity++ & 1)) ? 1 : 0;
    int n_small_plain = b->n_small_plain_tiles;
    if (unified) {
        for (int c = 0; c < kNumFast; c++) if (b->in_ulist[ul][c]) nt[c] = 0;
        if (b->small_in_ulist[ul]) n_small_plain = 0;
    }
    // The classes are independent (model.rs:102-180: surfaces never read what another surface wrote in the
    // same sub-timestep): spread them over the batch's stream and its side streams so that one class's
    // tail overlaps the next class's head.
    int n_launch = 0;
    for (int c = 0; c < kNumFast; c++) n_launch += nt[c] > 0;
    // small tiles with cavities: the streamed ones come first, the fused workgroups' after them
    int n_cav_tiles = streamed_only ? b->n_smallcav_stream_tiles : b->n_small_tiles - b->n_small_plain_tiles;
    if (unified && b->smallcav_in_ulist[ul]) n_cav_tiles = 0;
    n_launch += unified;
    n_launch += n_small_plain > 0;
    n_launch += n_cav_tiles > 0;
    n_launch += b->n_gen_tiles > b->n_small_tiles;
    static const bool no_fork = getenv("HEAT_AMD_NO_FORK") != nullptr;  // measurement: every class on the batch's stream
    // (not inside a stream capture: three of the fuzzer's cases — batches without palettes, whose two or three class launches
    // were forked onto side streams inside the captured sub-timestep — ended in a hang or a crash of the process inside the
    // HIP runtime at the SECOND capture of a batch (a march call of another length); captured, the launches follow each other
    // on the batch's stream, where a graph leaves no gap between them anyway)
    const bool fork = n_launch > 1 && b->side[0] != nullptr && !no_fork && !b->capturing;
    // Exactly the side streams that get work are forked, and every forked one is joined below: inside a stream capture a
    // side stream that waits for the fork event and is never joined leaves the capture unjoined — found by tools/fuzz.py
    // as a hang or a crash of the process some captures later (batches without palettes: two or three class launches on
    // three side streams).
    const int n_side = fork ? std::min(n_launch - 1, (int)heat_batch::kSideStreams) : 0;
    int slot = 0;
    auto next_stream = [&]() -> hipStream_t {
        if (n_side == 0) return b->stream;
        const int s = slot++ % (n_side + 1);
        return s == 0 ? b->stream : b->side[s - 1];
    };
    if (n_side > 0) {
        (void)hipEventRecord(b->ev_fork, b->stream);
        for (int i = 0; i < n_side; i++) (void)hipStreamWaitEvent(b->side[i], b->ev_fork, 0);
    }
    if (unified) {
        // the parts follow each other on ONE stream (a dependency between two streams costs more than the tail of a part)
        hipStream_t us = next_stream();
        // the light part first: its wavefronts are the latency-bound ones, the wide part's tail is the shorter
        static const int order_env = getenv("HEAT_AMD_STREAM_ORDER") ? atoi(getenv("HEAT_AMD_STREAM_ORDER")) : 0;  // measurement
        const int seq[2][kStreamVariants] = {{3, 1, 2, 0}, {0, 1, 2, 3}};
        for (int qi = 0; qi < kStreamVariants; qi++) {
            // (a reversed sweep takes the parts in the opposite order too, each from its end)
            const int q = seq[order_env ? 1 : 0][rev ? kStreamVariants - 1 - qi : qi];
            const int first = b->ulist_part[ul][q][0], n = b->ulist_part[ul][q][1];
            if (n > 0)
                launch_surfaces_stream(q, b->d_ulist[ul].p + first, n, b->na, b->gen_base, b->sa, b->d_weather.p, b->d_step.p,
                                       step_fixed, b->d_zone_T.p, b->d_flags.p,
                                       b->d_ucount.p + ul * b->ucount_stride + first, b->n_cu, us, rev);
        }
    }
    // work of each class in node slots, to size the persistent grids
    double work[kNumFast], total_work = 0.0;
    for (int c = 0; c < kNumFast; c++) {
        work[c] = (double)nt[c] * kFastM[c];
        total_work += work[c];
    }
    total_work += 2.0 * b->n_gen_tiles;  // small / general tiles: a few nodes per lane
    // biggest classes first
    int order[kNumFast];
    for (int c = 0; c < kNumFast; c++) order[c] = c;
    std::sort(order, order + kNumFast, [&](int x, int y) { return nt[x] > nt[y]; });
    for (int q = 0; q < kNumFast; q++) {
        const int c = order[q];
        if (nt[c] <= 0) continue;
        launch_surfaces_fast(kFastM[c], kFastNM[c] ? (b->class_has_chunks[c] ? 2 : 1) : 0, kFastPAL[c], kFastCAV[c], work[c] / total_work,
                             b->d_fast_tiles[c].p, nt[c], b->na,
                             b->sa, b->d_weather.p, b->d_step.p, step_fixed, b->d_zone_T.p, b->d_flags.p,
                             b->d_nomass_iters.p + b->nm_count_base[c], b->n_cu, next_stream(), rev);
    }
    unsigned long long *cnt = b->d_nomass_iters.p + b->nm_count_base[kNumFast];
    if (n_small_plain > 0)
        launch_surfaces_small(0, b->d_gen_tiles.p, n_small_plain, b->na, b->gen_base, b->sa, b->d_cavs.p,
                              b->d_weather.p, b->d_step.p, step_fixed, b->d_zone_T.p, b->d_flags.p, cnt,
                              next_stream());
    if (n_cav_tiles > 0)
        launch_surfaces_small(1, b->d_gen_tiles.p + b->n_small_plain_tiles, n_cav_tiles,
                              b->na, b->gen_base, b->sa, b->d_cavs.p, b->d_weather.p, b->d_step.p, step_fixed,
                              b->d_zone_T.p, b->d_flags.p, cnt + (size_t)b->n_small_plain_tiles * kWave,
                              next_stream());
    if (b->n_gen_tiles > b->n_small_tiles)
        launch_surfaces_general(b->d_gen_tiles.p + b->n_small_tiles, b->n_gen_tiles - b->n_small_tiles, b->na,
                                b->gen_base, b->sa, b->d_cavs.p, b->d_scratch.p, b->d_weather.p, b->d_step.p,
                                step_fixed, b->d_zone_T.p, b->d_flags.p, cnt + (size_t)b->n_small_tiles * kWave,
                                next_stream());
    for (int i = 0; i < n_side; i++) {
        (void)hipEventRecord(b->ev_join[i], b->side[i]);
        (void)hipStreamWaitEvent(b->stream, b->ev_join[i], 0);
    }
}

// mode 0 / 1 / 2 as k_zones; mode 3: as mode 0 (full update, advances the step counter) but only the zones no
// fused workgroup owns (the cluster-resident march balances its own zones in LDS).
void enqueue_zones(heat_batch *b, int mode) {
    const int32_t *zl = b->d_zlist.p;
    int nl = b->n_touched;
    if (mode == 3) { zl = b->d_stream_zones.p; nl = b->n_stream_zones; }
    if (mode == 5) { mode = 3; }  // sharded batch, nothing fused in this call: every zone it owns (d_zlist), full update
    if (mode == 4) { zl = b->d_zlist_stream.p; nl = b->n_touched_stream; mode = 2; }  // sharded, beside a fused march
    launch_zones(b->d_zone_off.p, b->d_zone_entries.p, b->d_zone_contrib.p, b->d_zone_a0.p, b->d_zone_b0.p,
                 b->d_zone_vol.p, b->d_zone_T.p, b->partial_ptr, (int)b->n_zones, b->dt, b->d_step.p,
                 b->d_flags.p, mode, zl, nl, b->d_slot_of.p, b->n_shared, b->zone_rows, b->stream);
}

// The cluster-resident march: every fused workgroup marches n_sub sub-timesteps in one launch per class.
// streamed_beside: other surfaces of the batch are streamed on the batch's stream while this launch runs.
int enqueue_fused(heat_batch *b, int n_sub, hipStream_t st, bool streamed_beside = false) {
    FusedArgs fa{};
    {   // zig-zag over march calls (see enqueue_surfaces): what the last call wrote back last is this call's first read
        static const bool zigzag_off = getenv("HEAT_AMD_NO_ZIGZAG") != nullptr;  // measurement
        fa.reverse = (!zigzag_off && (b->fused_parity++ & 1)) ? 1 : 0;
    }
    fa.zones = b->d_fzones.p;
    fa.zone_eoff = b->d_fzone_eoff.p;
    fa.slots = b->d_fslots.p;
    fa.side_area = b->d_side_area.p;
    fa.side_lzone = b->d_side_lzone.p;
    fa.a0 = b->d_zone_a0.p;
    fa.b0 = b->d_zone_b0.p;
    fa.vol = b->d_zone_vol.p;
    fa.zone_T = b->d_zone_T.p;
    fa.dt = b->dt;
    fa.n_sub = n_sub;
    fa.gen_tiles = b->d_gen_tiles.p;
    fa.gen_base = b->gen_base;
    fa.small_iters = b->d_nomass_iters.p + b->nm_count_base[kNumFast];
    // Beside an exchange loop (sharded batch with shared zones) the fused launch must not take every wavefront slot
    // of the chip: the loop's small kernels and the collective would each wait tens of microseconds for a slot
    // (measured: 27-74 us instead of 5) and become the critical path. So the launch holds a few workgroups fewer
    // than the chip has room for, and its workgroups take their FusedBlocks from a queue.
    static const int reserve_env = getenv("HEAT_AMD_FUSED_RESERVE") ? atoi(getenv("HEAT_AMD_FUSED_RESERVE")) : -1;
    const bool beside_exchange = (b->comm != nullptr && b->shared_set && b->n_shared > 0) || reserve_env >= 0 || streamed_beside;
    // share of the batch's padded nodes that is streamed beside this launch: that much of the chip is left to it
    double streamed_share = 0.0;
    if (streamed_beside) {
        double ns = 4.0 * kWave * (b->n_gen_tiles - (b->n_small_tiles - b->n_small_plain_tiles - b->n_smallcav_stream_tiles)), nf = 0.0;
        for (int c = 0; c < kNumFast; c++) {
            ns += (double)b->n_stream_tiles[c] * kFastM[c] * kWave;
            nf += (double)(b->n_fast_tiles[c] - b->n_stream_tiles[c]) * kFastM[c] * kWave;
        }
        streamed_share = ns / std::max(ns + nf, 1.0);
    }
    const int n_cu = b->n_cu;
    if (beside_exchange && b->d_fqueue.n == 0) HIP_TRY(b->d_fqueue.zeros(kNumFast * 4));
    for (int c = 0; c < kNumFast; c++)
        for (int g2 = 0; g2 < 4; g2++) {
            const int nb = (int)b->h_fblocks[c][g2].size();
            if (nb == 0) continue;
            fa.blocks = b->d_fblocks[c][g2].p;
            fa.n_blocks = nb;
            fa.queue = nullptr;
            int grid = nb;
            if (beside_exchange) {
                const int fw = (g2 & 1) ? 8 : 4;
                static const int room_env = getenv("HEAT_AMD_FUSED_ROOM") ? atoi(getenv("HEAT_AMD_FUSED_ROOM")) : 0;  // tests
                const int room = room_env > 0 ? room_env : n_cu * fused_blocks_per_cu(kFastM[c], kFastCAV[c], g2 >> 1, fw, b->na.pal_stride);
                const int reserve = reserve_env >= 0 ? reserve_env
                                                     : std::min(room / 2, std::max(fw == 4 ? 16 : 8, (int)(room * streamed_share)));
                if (nb > room - reserve) {
                    grid = std::max(1, room - reserve);
                    fa.queue = b->d_fqueue.p + c * 4 + g2;
                    HIP_TRY(hipMemsetAsync(fa.queue, 0, sizeof(unsigned int), st));
                }
            }
            static const bool trace = getenv("HEAT_AMD_TRACE") != nullptr;
            if (trace)
                fprintf(stderr, "heat_amd: fused launch%s: %d workgroups for %d blocks (class %d, list %d, %d sub-timesteps)\n",
                        fa.queue ? " on the work queue" : "", grid, nb, c, g2, n_sub);
            b->n_fused_launches++;
            // (a class whose walls hold no-mass chunks other than facings marches with the variant that carries the chunk loop)
            const int nm_variant = kFastNM[c] ? ((b->class_has_chunks[c] && !(g2 >> 1) && !kFastCAV[c] && kFastM[c] <= 8) ? 2 : 1) : 0;
            HIP_TRY(launch_surfaces_fused(kFastM[c], nm_variant, kFastCAV[c], g2 >> 1, (g2 & 1) ? 8 : 4, grid, b->d_fast_tiles[c].p, b->n_fast_tiles[c],
                                          b->na, b->sa, b->d_weather.p, b->d_flags.p,
                                          b->d_nomass_iters.p + b->nm_count_base[c], fa, st));
        }
    // Teams (clusters larger than a workgroup): n_teams x kTeamMax workgroups, ALL of them on the chip at once — a member
    // waits for the others' zone sums, so no workgroup of the launch may have to wait for a slot a waiting one holds.
    for (int c = 0; c < kNumFast; c++) {
        const int n_super = (int)b->h_team_supers[c].size();
        if (n_super == 0) continue;
        if (n_sub > 4095) return fail(HEAT_E_INVALID_ARG, "a march call of %d sub-timesteps: the team exchange tags 4095 at most", n_sub);
        // what the chip holds of this variant, less a margin (the hardware may admit a workgroup per compute unit fewer than
        // the occupancy arithmetic says); other kernels in flight only delay a member's start, they end by themselves
        // (at most two per compute unit whatever the query says: the variants hold 171-256 registers)
        static const int team_room_env = getenv("HEAT_AMD_TEAM_ROOM") ? atoi(getenv("HEAT_AMD_TEAM_ROOM")) : 0;  // measurement
        const int room = team_room_env > 0 ? team_room_env : n_cu * std::min(2, fused_team_blocks_per_cu(kFastM[c], kFastNM[c], b->na.pal_stride));
        int team_size = 2;
        for (const FusedSuper &su : b->h_team_supers[c]) team_size = std::max(team_size, (int)su.n_members);
        const int n_teams = std::min(n_super, std::max(0, room - room / 8) / team_size);
        if (n_teams >= 1 && (n_super + n_teams - 1) / n_teams > 1023)
            return fail(HEAT_E_SIZE, "%d clusters for %d teams: the exchange tags 1023 rounds per launch at most", n_super, n_teams);
        if (n_teams < 1) return fail(HEAT_E_DEVICE, "the device holds %d workgroups of the team variant: not one team", room);
        if (b->xbuf_teams < n_teams) {
            HIP_TRY(b->d_xbuf.zeros((size_t)n_teams * 2 * kTeamZones * kTeamMax * 4));
            b->xbuf_teams = n_teams;
        }
        b->team_epoch++;
        if ((b->team_epoch & 1023u) == 0) {  // the 10-bit launch number wraps: no granule of the launch 1024 before may survive
            HIP_TRY(hipMemsetAsync(b->d_xbuf.p, 0, b->d_xbuf.n * sizeof(unsigned long long), st));
            b->team_epoch++;
        }
        fa.blocks = b->d_team_blocks[c].p;
        fa.n_blocks = (int)b->h_team_blocks[c].size();
        fa.queue = nullptr;
        fa.supers = b->d_team_supers[c].p;
        fa.n_super = n_super;
        fa.team_zinfo = b->d_team_zinfo.p;
        fa.xbuf = b->d_xbuf.p;
        fa.tag_base = (b->team_epoch & 1023u) << 22;
        fa.team_size = team_size;
        static const bool trace = getenv("HEAT_AMD_TRACE") != nullptr;
        if (trace)
            fprintf(stderr, "heat_amd: team launch: %d teams of %d workgroups for %d clusters (class %d, %d sub-timesteps)\n",
                    n_teams, team_size, n_super, c, n_sub);
        b->n_fused_launches++;
        HIP_TRY(launch_surfaces_fused(kFastM[c], kFastNM[c], 0, 2, 4, n_teams * team_size, b->d_fast_tiles[c].p, b->n_fast_tiles[c], b->na,
                                      b->sa, b->d_weather.p, b->d_flags.p, b->d_nomass_iters.p + b->nm_count_base[c], fa, st));
    }
    return HEAT_OK;
}

hipEvent_t next_event(heat_batch *b) {
    if (b->ev_used == b->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        b->ev_pool.push_back(e);
    }
    return b->ev_pool[b->ev_used++];
}

// A march call that forks work onto the fused stream, or swaps the partials buffer, must leave the batch consistent
// on EVERY return path: an early error return would otherwise leave the side stream unjoined (a later download that
// synchronises only the batch's stream would race with the still-running launch) or the partials pointer swapped.
struct MarchGuard {
    heat_batch *b;
    double *saved_partial;
    bool forked = false, joined = false;
    explicit MarchGuard(heat_batch *b_) : b(b_), saved_partial(b_->partial_ptr) {}
    ~MarchGuard() {
        b->partial_ptr = saved_partial;
        if (forked && !joined && b->fused_stream) (void)hipStreamSynchronize(b->fused_stream);  // error path: wait it out
    }
};

}  // namespace

// ===========================================================================
extern "C" {


int heat_batch_create(const heat_batch_desc *desc, heat_batch **out) {
    heat_batch_options opt;
    memset(&opt, 0, sizeof opt);
    opt.device = -1;
    opt.n_ranks = 1;
    return heat_batch_create_ex(desc, &opt, out);
}

int heat_batch_create_ex(const heat_batch_desc *desc, const heat_batch_options *opt_in, heat_batch **out) {
    if (!out) return fail(HEAT_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    heat_batch_options opt;
    memset(&opt, 0, sizeof opt);
    opt.device = -1;
    opt.n_ranks = 1;
    if (opt_in) opt = *opt_in;
    if (opt.n_ranks < 1) opt.n_ranks = 1;
    if (opt.nodes_per_lane != 0 && opt.nodes_per_lane != 4 && opt.nodes_per_lane != 8 && opt.nodes_per_lane != 16)
        return fail(HEAT_E_INVALID_ARG, "nodes_per_lane must be 0, 4, 8 or 16");
    int rc;
    {
        std::string err;
        rc = check_desc(desc, err);
        if (rc) return fail(rc, "%s", err.c_str());
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(HEAT_E_DEVICE, "no HIP device available: this library has no CPU fallback");
    heat_batch *b = new heat_batch();
    if (opt.device >= 0) {
        b->device = opt.device;
    } else if (hipGetDevice(&b->device) != hipSuccess) {
        delete b;
        return fail(HEAT_E_DEVICE, "hipGetDevice failed");
    }
    b->n_ranks = opt.n_ranks;
    b->rank = opt.rank;
    b->use_graph = opt.use_graph != 0;
    b->fusion_on = opt.no_fusion != 1;
    rc = select_device(b);
    if (!rc) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, b->device) == hipSuccess && prop.multiProcessorCount > 0) b->n_cu = prop.multiProcessorCount;
    }
    if (!rc) {
        if (opt.stream) {
            b->stream = reinterpret_cast<hipStream_t>(opt.stream);
        } else {
            hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
            if (e != hipSuccess) rc = fail(HEAT_E_DEVICE, "hipStreamCreate failed: %s", hipGetErrorString(e));
            b->own_stream = true;
        }
    }
    if (!rc) {
        for (int i = 0; i < heat_batch::kSideStreams && !rc; i++) {
            if (hipStreamCreateWithFlags(&b->side[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&b->ev_join[i], hipEventDisableTiming) != hipSuccess)
                rc = fail(HEAT_E_DEVICE, "side stream creation failed");
        }
        if (!rc && hipStreamCreateWithFlags(&b->fused_stream, hipStreamNonBlocking) != hipSuccess)
            rc = fail(HEAT_E_DEVICE, "side stream creation failed");
        if (!rc && (hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&b->ev_fused, hipEventDisableTiming) != hipSuccess))
            rc = fail(HEAT_E_DEVICE, "event creation failed");
    }
    if (!rc) rc = build(b, desc, opt);
    if (rc) {
        delete b;
        return rc;
    }
    *out = b;
    return HEAT_OK;
}

int heat_batch_create_shard(const heat_batch_desc *desc, const heat_batch_options *opt, const int32_t *rank_of_surface,
                            heat_batch **out) {
    if (!desc || !opt || !rank_of_surface || !out) return fail(HEAT_E_INVALID_ARG, "NULL argument");
    {
        std::string err;
        const int rc = check_desc(desc, err);
        if (rc) return fail(rc, "%s", err.c_str());
    }
    for (int64_t s = 0; s < desc->n_surfaces; s++)
        if (rank_of_surface[s] < 0 || rank_of_surface[s] >= std::max(opt->n_ranks, 1))
            return fail(HEAT_E_SIZE, "surface %lld: rank %d outside [0, %d)", (long long)s, rank_of_surface[s], std::max(opt->n_ranks, 1));
    ShardDesc sh;
    sh.build(desc, rank_of_surface, opt->rank);
    int rc = heat_batch_create_ex(&sh.desc, opt, out);
    if (rc) return rc;
    heat_batch *b = *out;
    // Zones by the ranks that face them: a zone nobody faces falls to rank z % n_ranks (it still follows a0 / b0,
    // model.rs:410-423); when no zone is faced from two ranks the batch marches without any exchange.
    const int64_t Z = desc->n_zones;
    const int n_ranks = std::max(opt->n_ranks, 1);
    std::vector<int32_t> first(Z, -1);
    std::vector<uint8_t> shared(Z, 0), owned(Z, 0);
    auto touch = [&](int32_t z, int32_t rk) {
        if (first[z] < 0) first[z] = rk;
        else if (first[z] != rk) shared[z] = 1;
    };
    for (int64_t s = 0; s < desc->n_surfaces; s++) {
        if (desc->front_kind[s] == HEAT_BOUNDARY_SPACE) touch(desc->front_zone[s], rank_of_surface[s]);
        if (desc->back_kind[s] == HEAT_BOUNDARY_SPACE) touch(desc->back_zone[s], rank_of_surface[s]);
    }
    int64_t n_shared = 0;
    for (int64_t z = 0; z < Z; z++) {
        n_shared += shared[z];
        owned[z] = (first[z] < 0 && z % n_ranks == opt->rank) ? 1 : 0;
    }
    rc = heat_batch_set_owned_zones(b, owned.data());
    if (!rc && n_ranks > 1 && n_shared == 0) rc = heat_batch_set_shared_zones(b, nullptr, 0);
    if (rc) {
        heat_batch_destroy(b);
        *out = nullptr;
    }
    return rc;
}

void heat_batch_destroy(heat_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    delete b;
}

int64_t heat_batch_n_surfaces(const heat_batch *b) { return b ? b->n_surf : 0; }
int64_t heat_batch_n_nodes(const heat_batch *b) { return b ? b->n_nodes : 0; }
int64_t heat_batch_n_zones(const heat_batch *b) { return b ? b->n_zones : 0; }
int64_t heat_batch_algorithmic_bytes(const heat_batch *b) { return b ? b->algorithmic_bytes : 0; }
int heat_batch_class_counts(const heat_batch *b, int64_t counts[5]) {
    if (!b || !counts) return fail(HEAT_E_INVALID_ARG, "NULL argument");
    for (int i = 0; i < 5; i++) counts[i] = b->class_counts[i];
    return HEAT_OK;
}

static int transfer_in(heat_batch *b, const double *state, size_t n_state, bool full) {
    if (!b || !state) return fail(HEAT_E_INVALID_ARG, "NULL argument");
    if ((int64_t)n_state != b->n_state) return fail(HEAT_E_SIZE, "n_state %zu, batch was created for %lld", n_state, (long long)b->n_state);
    int rc = select_device(b);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipMemcpy(b->d_state.p, state, n_state * sizeof(double), hipMemcpyHostToDevice));
    if (full) {
        for (int c = 0; c < kNumFast; c++)
            launch_nodes_fast(kFastM[c], b->d_fast_tiles[c].p, b->n_fast_tiles[c], b->d_T.p, b->d_meta.p,
                              b->d_first_slot.p, b->d_state.p, 0, b->d_cls.p, b->stream);
        launch_nodes_general(b->d_gen_tiles.p, b->n_gen_tiles, b->d_T.p, b->d_meta.p, b->d_first_slot.p,
                             b->d_state.p, 0, b->stream);
    }
    launch_surf_scalars((int)b->n_surf, b->sl, b->d_side_dyn.p, b->d_side_out.p, b->d_side_alpha.p, b->d_state.p, 0,
                        full ? 3 : 1, b->stream);
    launch_zone_scalars((int)b->n_zones, b->d_zone_slot.p, b->d_zone_T.p, b->d_state.p, 0, b->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(b->stream));
    return HEAT_OK;
}

int heat_batch_upload_state(heat_batch *b, const double *state, size_t n_state) {
    const int rc = transfer_in(b, state, n_state, true);
    if (rc == HEAT_OK) b->host_zones_stale = false;
    return rc;
}
// Only what other modules write between two marches (surface_trait.rs:81-125's irradiance slots, the zones'
// dry-bulb slots): gathered on the host into pinned memory, one copy, converted on the device.
int heat_batch_upload_inputs(heat_batch *b, const double *state, size_t n_state) {
    if (!b || !state) return fail(HEAT_E_INVALID_ARG, "NULL argument");
    if ((int64_t)n_state != b->n_state) return fail(HEAT_E_SIZE, "n_state %zu, batch was created for %lld", n_state, (long long)b->n_state);
    int rc = select_device(b);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));  // (the staging buffer may still feed the previous call's copy)
    const int64_t S = b->n_surf, Z = b->n_zones;
    double *pin = b->h_pin;
    const int64_t *slots = b->h_in_slots.data();
    // gathered by the thread pool and copied in pieces: the copy of one piece travels while the next is gathered
    const int64_t total = 4 * S;
    const int n_pieces = total >= (int64_t)1 << 20 ? 4 : 1;
    for (int pc = 0; pc < n_pieces; pc++) {
        const int64_t a = total * pc / n_pieces, e = total * (pc + 1) / n_pieces;
        b->pool->run(e - a, [&](int64_t i0, int64_t i1) {
            for (int64_t i = a + i0; i < a + i1; i++) pin[i] = state[slots[i]];
        });
        if (e > a)
            HIP_TRY(hipMemcpyAsync(b->d_compact.p + a, pin + a, (size_t)(e - a) * sizeof(double), hipMemcpyHostToDevice, b->stream));
    }
    // The zones' dry-bulb slots are taken from the caller's state only while that state holds what this path last
    // computed: after a march whose outputs left HEAT_OUT_ZONE_TEMPERATURES out (heat_batch_march_ex, or a resident march
    // without a download) the caller's zone slots are OLDER than the device's — taking them would set the zones back by a
    // call (found by tools/fuzz.py). heat_batch_upload_state always takes everything.
    const bool take_zones = Z > 0 && !b->host_zones_stale;
    if (take_zones) {
        for (int64_t z = 0; z < Z; z++) pin[4 * S + z] = state[b->h_zone_slot_h[z]];
        HIP_TRY(hipMemcpyAsync(b->d_compact.p + 4 * S, pin + 4 * S, (size_t)Z * sizeof(double), hipMemcpyHostToDevice, b->stream));
    }
    launch_inputs_compact((int)S, take_zones ? (int)Z : 0, b->d_compact.p, b->d_side_alpha.p, b->d_side_dyn.p, b->d_zone_T.p, b->sl,
                          b->direct_runs.empty() ? nullptr : b->d_state.p, b->stream);
    HIP_TRY(hipGetLastError());
    return HEAT_OK;
}

int heat_batch_download_state(heat_batch *b, double *state, size_t n_state) {
    return heat_batch_download_outputs(b, state, n_state, HEAT_OUT_ALL);
}

// The outputs of this path into the caller's state: gathered on the device into a compact buffer in the caller's
// surface order, copied through two pinned staging halves (the copy of one piece runs while the thread pool
// scatters the piece before it), only the slots the path owns are written.
static int download_impl(heat_batch *b, double *state, size_t n_state, int32_t what, bool inputs_fresh);

int heat_batch_download_outputs(heat_batch *b, double *state, size_t n_state, int32_t what) {
    return download_impl(b, state, n_state, what, false);
}

// inputs_fresh: the irradiance slots of `state` were uploaded from this very array in the same call (heat_batch_march):
// the mirror then equals the caller's state on every slot of the direct runs that is not an output, and whole runs can
// be copied back.
static int download_impl(heat_batch *b, double *state, size_t n_state, int32_t what, bool inputs_fresh) {
    if (!b || !state) return fail(HEAT_E_INVALID_ARG, "NULL argument");
    if ((int64_t)n_state != b->n_state) return fail(HEAT_E_SIZE, "n_state %zu, batch was created for %lld", n_state, (long long)b->n_state);
    int rc = select_device(b);
    if (rc) return rc;
    const int64_t S = b->n_surf, Z = b->n_zones, N = b->n_nodes;
    const bool delivers_zones = (what & HEAT_OUT_ZONE_TEMPERATURES) != 0;
    bool nodes = (what & HEAT_OUT_NODE_TEMPERATURES) != 0;
    bool scalars = (what & (HEAT_OUT_SURFACE_SCALARS | HEAT_OUT_ZONE_TEMPERATURES)) != 0;
    if (nodes && (what & HEAT_OUT_SURFACE_SCALARS) && inputs_fresh && !b->direct_runs.empty()) {
        // Direct: node temperatures, hs and heat flows go into the mirror; its runs — outputs and the inputs this call
        // uploaded, nothing else — are copied straight into the caller's array (pageable or not: 56 GB/s either way on
        // this platform, tools/pcie_bw.py). The zones follow through the staged path below.
        for (int c = 0; c < kNumFast; c++)
            launch_nodes_fast(kFastM[c], b->d_fast_tiles[c].p, b->n_fast_tiles[c], b->d_T.p, b->d_meta.p,
                              b->d_first_slot.p, b->d_state.p, 1, b->d_cls.p, b->stream);
        launch_nodes_general(b->d_gen_tiles.p, b->n_gen_tiles, b->d_T.p, b->d_meta.p, b->d_first_slot.p, b->d_state.p, 1,
                             b->stream);
        launch_surf_scalars((int)S, b->sl, b->d_side_dyn.p, b->d_side_out.p, b->d_side_alpha.p, b->d_state.p, 1, 2, b->stream);
        HIP_TRY(hipGetLastError());
        for (auto &r : b->direct_runs)
            HIP_TRY(hipMemcpyAsync(state + r.first, b->d_state.p + r.first, (size_t)(r.second - r.first) * sizeof(double),
                                   hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        nodes = false;
        what &= ~(HEAT_OUT_NODE_TEMPERATURES | HEAT_OUT_SURFACE_SCALARS);
        scalars = (what & HEAT_OUT_ZONE_TEMPERATURES) != 0;
        if (!scalars) return HEAT_OK;
    }
    if (nodes) {
        for (int c = 0; c < kNumFast; c++)
            launch_nodes_fast(kFastM[c], b->d_fast_tiles[c].p, b->n_fast_tiles[c], b->d_T.p, b->d_meta.p,
                              b->d_compact_off.p, b->d_compact.p, 1, b->d_cls.p, b->stream);
        launch_nodes_general(b->d_gen_tiles.p, b->n_gen_tiles, b->d_T.p, b->d_meta.p, b->d_compact_off.p, b->d_compact.p,
                             1, b->stream);
    }
    if (scalars)
        launch_outputs_compact((int)S, (int)Z, b->d_side_out.p, b->d_orig_of32.p, b->d_zone_T.p, b->d_compact.p + N, b->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(b->stream));
    const size_t half = b->pin_doubles / 2;
    // the scalar part first (one piece), then the node pieces; piece i uses staging half i % 2
    struct Piece { int64_t off, count, s0, s1; bool is_nodes; };
    std::vector<Piece> pieces;
    // (zones alone: only the tail of the scalar part travels)
    const int64_t sc_off = (what & HEAT_OUT_SURFACE_SCALARS) ? 0 : 4 * S;
    if (scalars) pieces.push_back({N + sc_off, 4 * S + Z - sc_off, 0, S, false});
    if (nodes)
        for (auto &ch : b->out_chunks)
            pieces.push_back({b->h_node_off[ch.first], b->h_node_off[ch.second] - b->h_node_off[ch.first], ch.first, ch.second, true});
    auto issue = [&](size_t i) -> hipError_t {
        const Piece &p = pieces[i];
        // (the scalar piece may be longer than a half: it is alone in the buffer then — the node pieces start after it)
        double *dst = b->h_pin + (i % 2) * half;
        hipError_t e = hipMemcpyAsync(dst, b->d_compact.p + p.off, (size_t)p.count * sizeof(double), hipMemcpyDeviceToHost, b->copy_stream);
        if (e != hipSuccess) return e;
        return hipEventRecord(b->ev_copy[i % 2], b->copy_stream);
    };
    const bool big_scalars = scalars && (size_t)(4 * S + Z) > half;
    for (size_t i = 0; i < pieces.size(); i++) {
        if (i == 0) HIP_TRY(issue(0));
        HIP_TRY(hipEventSynchronize(b->ev_copy[i % 2]));
        // the next copy runs while this piece is scattered (not into a buffer the scalar piece spills over)
        const bool overlap = i + 1 < pieces.size() && !(big_scalars && i == 0);
        if (overlap) HIP_TRY(issue(i + 1));
        const Piece &p = pieces[i];
        const double *src = b->h_pin + (i % 2) * half;
        if (p.is_nodes) {
            const int64_t base = p.off;
            b->pool->run(p.s1 - p.s0, [&](int64_t q0, int64_t q1) {
                for (int64_t s = p.s0 + q0; s < p.s0 + q1; s++)
                    memcpy(state + b->h_first_slot[s], src + (b->h_node_off[s] - base), (size_t)b->h_node_count[s] * sizeof(double));
            });
        } else {
            if (what & HEAT_OUT_SURFACE_SCALARS)
                b->pool->run(S, [&](int64_t s0, int64_t s1) {
                    for (int64_t s = s0; s < s1; s++) {
                        const double *v = src + 4 * s;  // hs_f, hs_b, flow_f, flow_b
                        state[b->h_out_slots[0][s]] = v[0];
                        state[b->h_out_slots[1][s]] = v[1];
                        state[b->h_out_slots[2][s]] = v[2];
                        state[b->h_out_slots[3][s]] = v[3];
                    }
                });
            if (what & HEAT_OUT_ZONE_TEMPERATURES)
                for (int64_t z = 0; z < Z; z++)
                    if (b->h_owned[z]) state[b->h_zone_slot_h[z]] = src[4 * S - sc_off + z];
        }
        if (!overlap && i + 1 < pieces.size()) HIP_TRY(issue(i + 1));
    }
    if (delivers_zones) b->host_zones_stale = false;  // (the caller's zone slots are the device's again)
    return HEAT_OK;
}

int heat_batch_set_weather(heat_batch *b, const heat_weather *weather, int32_t n_sub, const double *zone_a0,
                           const double *zone_b0) {
    if (!b || (!weather && n_sub > 0)) return fail(HEAT_E_INVALID_ARG, "NULL argument");
    if (n_sub < 0) return fail(HEAT_E_INVALID_ARG, "n_sub < 0");
    int rc = select_device(b);
    if (rc) return rc;
    // The pinned staging buffers are free again once the previous call's copies have run — they sit at the head of
    // that call's work, so this does not wait for the march itself: consecutive marches queue up back to back.
    if (b->staged) HIP_TRY(hipEventSynchronize(b->ev_staged));
    if ((size_t)n_sub > b->weather_cap) {
        HIP_TRY(hipStreamSynchronize(b->stream));  // kernels in flight read the device array that is about to go
        if (b->h_weather) HIP_TRY(hipHostFree(b->h_weather));
        b->h_weather = nullptr;
        const size_t cap = std::max<size_t>((size_t)n_sub, 64);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_weather), cap * sizeof(StepWeather)));
        HIP_TRY(b->d_weather.alloc(cap));
        b->weather_cap = cap;
        if (b->graph_exec) {  // the captured graph holds the old pointer
            (void)hipGraphExecDestroy(b->graph_exec);
            b->graph_exec = nullptr;
        }
    }
    for (int i = 0; i < n_sub; i++) {
        // sin/cos of the wind direction as is_windward takes them (surface.rs:40)
        b->h_weather[i] = StepWeather{weather[i].dry_bulb, std::sqrt(weather[i].wind_speed),
                                      std::sin(weather[i].wind_direction), std::cos(weather[i].wind_direction)};
    }
    b->n_weather = n_sub;
    const int64_t Z = b->n_zones;
    for (int64_t z = 0; z < Z; z++) {
        b->h_zone_ab[z] = zone_a0 ? zone_a0[z] : 0.0;
        b->h_zone_ab[Z + z] = zone_b0 ? zone_b0[z] : 0.0;
    }
    // weather, a0, b0 and the sub-timestep counter in one launch that reads the pinned buffers itself
    launch_begin_march(b->h_weather, b->d_weather.p, n_sub, b->h_zone_ab, b->d_zone_a0.p, b->d_zone_b0.p, (int)Z, b->d_step.p,
                       b->stream);
    if (!b->ev_staged) HIP_TRY(hipEventCreateWithFlags(&b->ev_staged, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->ev_staged, b->stream));
    b->staged = true;
    HIP_TRY(hipGetLastError());
    return HEAT_OK;
}

int heat_batch_step_surfaces(heat_batch *b, int32_t sub_step) {
    if (!b) return fail(HEAT_E_INVALID_ARG, "NULL batch");
    if (sub_step < 0 || sub_step >= b->n_weather) return fail(HEAT_E_INVALID_ARG, "sub_step %d outside the weather set (%d)", sub_step, b->n_weather);
    int rc = select_device(b);
    if (rc) return rc;
    b->host_zones_stale = true;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (b->timing) {
        e0 = next_event(b); e1 = next_event(b); e2 = next_event(b);
        if (!e0 || !e1 || !e2) return fail(HEAT_E_DEVICE, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(e0, b->stream));
    }
    enqueue_surfaces(b, sub_step);
    if (b->timing) HIP_TRY(hipEventRecord(e1, b->stream));
    enqueue_zones(b, b->shared_set ? 2 : 1);  // partial (a, b) of this rank's surfaces
    if (b->timing) {
        HIP_TRY(hipEventRecord(e2, b->stream));
        b->ev_triples.push_back({e0, e1, e2});
    }
    HIP_TRY(hipGetLastError());
    return HEAT_OK;
}

double *heat_batch_zone_partials(heat_batch *b) { return b ? b->partial_ptr : nullptr; }

int heat_batch_use_partials(heat_batch *b, double *partials_dev) {
    if (!b) return fail(HEAT_E_INVALID_ARG, "NULL batch");
    b->partial_ptr = partials_dev ? partials_dev : b->d_partial.p;
    return HEAT_OK;
}

int heat_batch_touched_zones(const heat_batch *b, uint8_t *mask) {
    if (!b || (!mask && b->n_zones > 0)) return fail(HEAT_E_INVALID_ARG, "NULL argument");
    for (int64_t z = 0; z < b->n_zones; z++) mask[z] = b->h_touched[z];
    return HEAT_OK;
}

int heat_batch_set_shared_zones(heat_batch *b, const int32_t *shared_zone, int32_t n_shared) {
    static const int32_t none = 0;
    if (n_shared == 0 && !shared_zone) shared_zone = &none;
    if (!b || n_shared < 0 || !shared_zone) return fail(HEAT_E_INVALID_ARG, "bad argument");
    int rc = select_device(b);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    std::vector<int32_t> slot(b->n_zones, -1), zl, sz(shared_zone, shared_zone + n_shared);
    for (int32_t i = 0; i < n_shared; i++) {
        if (sz[i] < 0 || sz[i] >= b->n_zones) return fail(HEAT_E_SIZE, "shared zone %d out of range", sz[i]);
        slot[sz[i]] = i;
    }
    for (int64_t z = 0; z < b->n_zones; z++) if (b->h_owned[z]) zl.push_back((int32_t)z);
    HIP_TRY(b->d_slot_of.upload(slot));
    HIP_TRY(b->d_zlist.upload(zl));
    HIP_TRY(b->d_shared_zone.upload(sz));
    b->n_touched = (int)zl.size();
    b->n_shared = n_shared;
    b->shared_set = true;
    if (b->comm) HIP_TRY(b->d_gathered.zeros((size_t)b->n_ranks * 2 * std::max(n_shared, 1)));
    if (b->partial_ptr == b->d_partial.p && (size_t)2 * n_shared > b->d_partial.n)
        return fail(HEAT_E_SIZE, "partials buffer too small");

    // ---- cluster-resident march: demote the workgroups that own a shared zone (from the create-time plan) ----
    // A demoted workgroup's tiles move, as descriptors, behind the streamed tiles of their class; the remaining
    // workgroups keep their slot lists (those are relative to the block's first tile).
    b->any_fused = false;
    int64_t n_fused_now = 0;
    std::vector<uint8_t> zone_fused(b->n_zones, 0);
    auto owns_shared_zone = [&](const FusedBlock &fb) {
        for (int j = 0; j < fb.n_zones; j++)
            if (slot[b->h_fzones[fb.first_zone + j]] >= 0) return true;
        return false;
    };
    // small-surface tiles of the workgroups (general layout): [plain | streamed with cavities | DEMOTED | kept | catch-all]
    std::vector<int32_t> new_first_small(b->h_gen_tiles0.size() + 1, 0);
    {
        const std::vector<GeneralTile> &g0 = b->h_gen_tiles0;
        const int head = b->n_small_plain_tiles + b->n_smallcav_stream_tiles0;
        std::vector<GeneralTile> g(g0.begin(), g0.begin() + head);
        int n_demoted = 0;
        for (int pass = 0; pass < 2; pass++)
            for (int c = 0; c < kNumFast; c++)
                for (int g2 = 0; g2 < 4; g2++)
                    for (const FusedBlock &fb : b->h_fblocks0[c][g2]) {
                        if (fb.n_small <= 0 || (pass == 0) != owns_shared_zone(fb)) continue;
                        new_first_small[fb.first_small] = (int32_t)g.size();
                        g.insert(g.end(), g0.begin() + fb.first_small, g0.begin() + fb.first_small + fb.n_small);
                        if (pass == 0) n_demoted += fb.n_small;
                    }
        g.insert(g.end(), g0.begin() + b->n_small_tiles, g0.end());
        b->n_smallcav_stream_tiles = b->n_smallcav_stream_tiles0 + n_demoted;
        HIP_TRY(b->d_gen_tiles.upload(g));
        b->h_gen_tiles_cur = g;
    }
    // (teams are never demoted: a zone shared with another rank cannot also be exchanged inside a team)
    for (int c = 0; c < kNumFast; c++)
        for (const FusedBlock &fb : b->h_team_blocks0[c])
            if (owns_shared_zone(fb))
                return fail(HEAT_E_INVALID_ARG, "zone shared between ranks belongs to a cluster marched by a team of workgroups "
                                                "(create the batch with no_fusion = 1, or do not share that zone)");
    for (int c = 0; c < kNumFast; c++) {
        const std::vector<FastTile> &t0 = b->h_tiles0[c];
        bool any = !b->h_team_blocks0[c].empty();
        for (int g2 = 0; g2 < 4; g2++) any = any || !b->h_fblocks0[c][g2].empty();
        if (!any) continue;
        std::vector<FastTile> t(t0.begin(), t0.begin() + b->n_stream_tiles0[c]);
        std::vector<FusedBlock> keep[4];
        for (int pass = 0; pass < 2; pass++)          // pass 0: demoted blocks' tiles, pass 1: kept blocks' tiles
            for (int g2 = 0; g2 < 4; g2++)
                for (const FusedBlock &fb : b->h_fblocks0[c][g2]) {
                    if ((pass == 0) != owns_shared_zone(fb)) continue;
                    FusedBlock nb = fb;
                    nb.first_tile = (int32_t)t.size();
                    if (fb.n_small > 0) nb.first_small = new_first_small[fb.first_small];
                    t.insert(t.end(), t0.begin() + fb.first_tile, t0.begin() + fb.first_tile + fb.n_tiles);
                    if (pass == 1) {
                        keep[g2].push_back(nb);
                        for (int j = 0; j < fb.n_zones; j++) zone_fused[b->h_fzones[fb.first_zone + j]] = 1;
                        for (int q = 0; q < fb.n_small; q++) n_fused_now += b->h_gen_tiles0[fb.first_small + q].G;
                    }
                }
        int n_kept_tiles = 0;
        {   // the teams' tiles stay where they are in the order of things: behind the kept workgroups' tiles
            std::vector<FusedBlock> tb = b->h_team_blocks0[c];
            for (FusedBlock &fb : tb) {
                const int first = (int)t.size();
                t.insert(t.end(), t0.begin() + fb.first_tile, t0.begin() + fb.first_tile + fb.n_tiles);
                fb.first_tile = first;
                n_kept_tiles += fb.n_tiles;
                for (int j = 0; j < fb.n_zones; j++) zone_fused[b->h_fzones[fb.first_zone + j]] = 1;
                for (int q = 0; q < fb.n_tiles; q++) n_fused_now += t[fb.first_tile + q].G;
            }
            b->h_team_blocks[c] = tb;
            HIP_TRY(b->d_team_blocks[c].upload(tb));
            b->any_fused = b->any_fused || !tb.empty();
        }
        for (int g2 = 0; g2 < 4; g2++) {
            for (const FusedBlock &fb : keep[g2]) {
                n_kept_tiles += fb.n_tiles;
                for (int q = 0; q < fb.n_tiles; q++) n_fused_now += t[fb.first_tile + q].G;
            }
            b->h_fblocks[c][g2] = keep[g2];
            HIP_TRY(b->d_fblocks[c][g2].upload(keep[g2]));
            b->any_fused = b->any_fused || !keep[g2].empty();
        }
        b->n_stream_tiles[c] = (int)t.size() - n_kept_tiles;
        HIP_TRY(b->d_fast_tiles[c].upload(t));
        b->h_tiles_cur[c] = t;
    }
    {
        std::vector<int32_t> szl, tz;
        for (int64_t z = 0; z < b->n_zones; z++) {
            b->h_zone_block[z] = zone_fused[z] ? 0 : -1;
            if (!zone_fused[z] && b->h_owned[z]) szl.push_back((int32_t)z);
        }
        b->n_stream_zones = (int)szl.size();
        HIP_TRY(b->d_stream_zones.upload(szl));
        for (int32_t z : zl) if (!zone_fused[z]) tz.push_back(z);
        b->n_touched_stream = (int)tz.size();
        HIP_TRY(b->d_zlist_stream.upload(tz));
    }
    b->n_fused_surfaces = n_fused_now;
    rc = rebuild_unified(b);
    if (rc) return rc;
    if (b->graph_exec) {  // the captured sub-timestep graph holds the old tile counts
        HIP_TRY(hipStreamSynchronize(b->stream));
        (void)hipGraphExecDestroy(b->graph_exec);
        b->graph_exec = nullptr;
    }
    HIP_TRY(hipDeviceSynchronize());  // (the uploads above went through the null stream; the batch's streams do not wait for it)
    return HEAT_OK;
}

int heat_comm_available(void) {
    if (rccl()) return HEAT_OK;
    return fail(HEAT_E_COMM, "cannot load librccl.so.1: %s", rccl_error().c_str());
}

int heat_comm_unique_id(uint8_t id[HEAT_COMM_ID_BYTES]) {
    if (!id) return fail(HEAT_E_INVALID_ARG, "NULL argument");
    Rccl *r = rccl();
    if (!r) return fail(HEAT_E_COMM, "cannot load librccl.so.1: %s", rccl_error().c_str());
    static_assert(sizeof(ncclUniqueId) == HEAT_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    RCCL_TRY(r, r->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return HEAT_OK;
}

int heat_batch_set_owned_zones(heat_batch *b, const uint8_t *owned) {
    if (!b || (!owned && b->n_zones > 0)) return fail(HEAT_E_INVALID_ARG, "NULL argument");
    for (int64_t z = 0; z < b->n_zones; z++) b->h_owned[z] = (owned[z] || b->h_touched[z]) ? 1 : 0;
    if (b->shared_set) {  // the zone lists follow
        std::vector<int32_t> sz;
        int rc = select_device(b);
        if (rc) return rc;
        HIP_TRY(hipMemcpy((sz.resize(b->n_shared), sz.data()), b->d_shared_zone.p, (size_t)b->n_shared * sizeof(int32_t),
                          hipMemcpyDeviceToHost));
        return heat_batch_set_shared_zones(b, sz.data(), b->n_shared);
    }
    std::vector<int32_t> szl;
    for (int64_t z = 0; z < b->n_zones; z++)
        if (b->h_zone_block[z] < 0 && b->h_owned[z]) szl.push_back((int32_t)z);
    b->n_stream_zones = (int)szl.size();
    HIP_TRY(b->d_stream_zones.upload(szl));
    return HEAT_OK;
}

int heat_batch_comm_init(heat_batch *b, const uint8_t id[HEAT_COMM_ID_BYTES]) {
    return heat_batch_comm_init_ex(b, id, nullptr, 0);
}

int heat_batch_comm_init_ex(heat_batch *b, const uint8_t id[HEAT_COMM_ID_BYTES], const int32_t *extra_shared,
                            int32_t n_extra) {
    if (!b || !id || n_extra < 0 || (n_extra > 0 && !extra_shared)) return fail(HEAT_E_INVALID_ARG, "bad argument");
    if (b->comm) return fail(HEAT_E_INVALID_ARG, "the batch already has a communicator");
    const int64_t Z = b->n_zones;
    // (checked before anything collective happens: a bad list must not leave this rank outside a communicator the
    // other ranks have already agreed on)
    for (int32_t i = 0; i < n_extra; i++)
        if (extra_shared[i] < 0 || extra_shared[i] >= Z) return fail(HEAT_E_SIZE, "shared zone %d out of range", extra_shared[i]);
    Rccl *r = rccl();
    if (!r) return fail(HEAT_E_COMM, "cannot load librccl.so.1: %s", rccl_error().c_str());
    int rc = select_device(b);
    if (rc) return rc;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    RCCL_TRY(r, r->CommInitRank(&b->comm, b->n_ranks, u, b->rank));
    // From here on a failure returns the batch to "sharded batch without a communicator": the communicator goes, so
    // that a retry is possible and a march cannot take the no-exchange path on zones other ranks share.
    const std::vector<uint8_t> owned_before = b->h_owned;
    auto agree = [&]() -> int {
        // agree on the zones more than one rank touches: sum of the touched masks
        std::vector<int32_t> mask(std::max<int64_t>(Z, 1), 0);
        for (int64_t z = 0; z < Z; z++) mask[z] = b->h_touched[z];
        DevBuf<int32_t> d_mask;
        HIP_TRY(d_mask.upload(mask));
        RCCL_TRY(r, r->AllReduce(d_mask.p, d_mask.p, mask.size(), ncclInt32, ncclSum, b->comm, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        HIP_TRY(hipMemcpy(mask.data(), d_mask.p, mask.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<uint8_t> is_shared(std::max<int64_t>(Z, 1), 0);
        for (int64_t z = 0; z < Z; z++) {
            is_shared[z] = mask[z] >= 2;
            // a zone no rank faces is finished by rank z % n_ranks (it still follows a0 / b0: model.rs:410-423)
            b->h_owned[z] = (b->h_touched[z] || (mask[z] == 0 && z % b->n_ranks == b->rank)) ? 1 : 0;
        }
        for (int32_t i = 0; i < n_extra; i++) is_shared[extra_shared[i]] = 1;  // (tests, rehearsals of the exchange)
        std::vector<int32_t> shared;
        for (int64_t z = 0; z < Z; z++) if (is_shared[z]) shared.push_back((int32_t)z);
        return heat_batch_set_shared_zones(b, shared.data(), (int32_t)shared.size());
    };
    rc = agree();
    if (rc) {
        const std::string why = heat::last_error();
        (void)r->CommDestroy(b->comm);
        b->comm = nullptr;
        b->shared_set = false;
        b->n_shared = 0;
        b->h_owned = owned_before;
        return fail(rc, "%s", why.c_str());
    }
    return HEAT_OK;
}

int32_t heat_batch_n_shared_zones(const heat_batch *b) { return (b && b->shared_set) ? b->n_shared : 0; }

int32_t heat_batch_comm_ranks(const heat_batch *b) { return (b && b->comm) ? b->n_ranks : 0; }

int heat_batch_comm_destroy(heat_batch *b) {
    if (!b) return fail(HEAT_E_INVALID_ARG, "NULL batch");
    if (!b->comm) return HEAT_OK;
    int rc = select_device(b);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    Rccl *r = rccl();
    if (r) (void)r->CommDestroy(b->comm);
    b->comm = nullptr;
    // back to "sharded batch without a communicator": the caller agrees on the shared zones again by its own means
    // (heat_batch_set_owned_zones / heat_batch_set_shared_zones) or gives the batch a new communicator
    b->shared_set = false;
    b->n_shared = 0;
    b->d_gathered.release();
    return HEAT_OK;
}

int heat_batch_step_zones(heat_batch *b, const double *gathered_dev, int32_t n_blocks) {
    if (!b) return fail(HEAT_E_INVALID_ARG, "NULL batch");
    if (n_blocks < 1) return fail(HEAT_E_INVALID_ARG, "n_blocks < 1");
    int rc = select_device(b);
    if (rc) return rc;
    const double *g = gathered_dev ? gathered_dev : b->partial_ptr;
    if (b->shared_set)
        launch_zone_update_shared(g, n_blocks, b->d_shared_zone.p, b->n_shared, b->d_zone_a0.p, b->d_zone_b0.p,
                                  b->d_zone_vol.p, b->d_zone_T.p, b->dt, b->d_flags.p, b->stream);
    else
        launch_zone_update(g, n_blocks, b->d_zone_a0.p, b->d_zone_b0.p, b->d_zone_vol.p, b->d_zone_T.p,
                           (int)b->n_zones, b->dt, b->d_step.p, b->d_flags.p, b->stream);
    HIP_TRY(hipGetLastError());
    return HEAT_OK;
}

int heat_batch_march_resident(heat_batch *b, const heat_weather *weather, int32_t n_sub, const double *zone_a0,
                              const double *zone_b0) {
    if (!b) return fail(HEAT_E_INVALID_ARG, "NULL batch");
    // A sharded batch needs a collective only for zones it shares with another rank: a partition along the clusters
    // (heat_partition) shares none, and every rank then marches on its own.
    const bool exchange = b->shared_set && b->n_shared > 0;
    if (b->n_ranks > 1 && !b->comm && (exchange || !b->shared_set))
        return fail(HEAT_E_INVALID_ARG, "sharded batch without a communicator: call heat_batch_comm_init, or drive it "
                                        "with heat_batch_step_surfaces / heat_batch_step_zones and your own collective "
                                        "(a batch that shares no zone — heat_batch_set_shared_zones(b, NULL, 0) — needs neither)");
    int rc = heat_batch_set_weather(b, weather, n_sub, zone_a0, zone_b0);
    if (rc) return rc;
    // A march of no sub-timestep (model.rs:369: the loop body never runs) is the head kernel and nothing else: the
    // graph branch below must never capture or replay an empty graph.
    if (n_sub == 0) return HEAT_OK;
    b->host_zones_stale = true;  // (until a download delivers the zones again)
    if (b->comm && exchange) {
        // Sharded sub-timestep, everything in order on the batch's stream (no cross-queue dependency anywhere):
        // this rank's surfaces -> zones only this rank touches finished, partial (a, b) of the shared zones ->
        // RCCL all-gather of the [2][n_shared] blocks over xGMI -> shared zones updated from the blocks, summed
        // in rank order (identical bits on every rank).
        Rccl *r = rccl();
        if (!r) return fail(HEAT_E_COMM, "RCCL is not loaded");
        MarchGuard guard(b);
        b->partial_ptr = b->d_partial.p;
        // The clusters that own no shared zone march beside the exchange loop, in one launch per class on a side
        // stream: only the few surfaces around the shared zones go through kernel -> all-gather -> kernel every
        // sub-timestep, and that chain is shorter than the fused march it runs next to.
        const bool fused = b->any_fused && b->fusion_on && n_sub >= (b->n_surf <= 8192 ? 1 : kFusedMinSubsteps) &&
                           b->fused_stream != nullptr;
        if (fused) {
            hipStream_t fs = b->fused_stream;
            HIP_TRY(hipEventRecord(b->ev_fork, b->stream));
            HIP_TRY(hipStreamWaitEvent(fs, b->ev_fork, 0));
            guard.forked = true;
            hipEvent_t f0 = nullptr, f1 = nullptr;
            if (b->timing) {
                f0 = next_event(b); f1 = next_event(b);
                if (!f0 || !f1) return fail(HEAT_E_DEVICE, "hipEventCreate failed");
                HIP_TRY(hipEventRecord(f0, fs));
            }
            rc = enqueue_fused(b, n_sub, fs);
            if (rc) return rc;
            if (b->timing) {
                HIP_TRY(hipEventRecord(f1, fs));
                b->ev_fused_pairs.push_back({f0, f1, n_sub});
            }
            HIP_TRY(hipEventRecord(b->ev_fused, fs));
        }
        for (int i = 0; i < n_sub; i++) {
            hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
            if (b->timing) {
                e0 = next_event(b); e1 = next_event(b); e2 = next_event(b);
                if (!e0 || !e1 || !e2) return fail(HEAT_E_DEVICE, "hipEventCreate failed");
                HIP_TRY(hipEventRecord(e0, b->stream));
            }
            enqueue_surfaces(b, i, fused);
            if (b->timing) HIP_TRY(hipEventRecord(e1, b->stream));
            enqueue_zones(b, fused ? 4 : 2);
            if (b->n_shared > 0) {
                RCCL_TRY(r, r->AllGather(b->d_partial.p, b->d_gathered.p, (size_t)2 * b->n_shared, ncclDouble, b->comm,
                                         b->stream));
                launch_zone_update_shared(b->d_gathered.p, b->n_ranks, b->d_shared_zone.p, b->n_shared, b->d_zone_a0.p,
                                          b->d_zone_b0.p, b->d_zone_vol.p, b->d_zone_T.p, b->dt, b->d_flags.p, b->stream);
            }
            if (b->timing) {
                HIP_TRY(hipEventRecord(e2, b->stream));
                b->ev_triples.push_back({e0, e1, e2});
            }
        }
        if (fused) {
            HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_fused, 0));
            guard.joined = true;
        }
        HIP_TRY(hipGetLastError());
        return HEAT_OK;
    }
    // Cluster-resident march: the fused workgroups march all n_sub sub-timesteps in one launch per class (on a
    // side stream when other surfaces are streamed beside them); whatever is not fused is streamed as before.
    // (a resident call costs about 150 us + 59 us per sub-timestep at 1 M x 32 — round 3, tools/short_calls.py: 262 / 144 / 108 /
    // 88 us per sub-timestep at 1 / 2 / 3 / 5 per call against 206 for the same batch streamed: resident from two on)
    const bool fused = b->any_fused && b->fusion_on && n_sub >= (b->n_surf <= 8192 ? 1 : kFusedMinSubsteps);
    bool streamed = !fused;
    if (fused) {
        for (int c = 0; c < kNumFast; c++) streamed = streamed || b->n_stream_tiles[c] > 0;
        streamed = streamed || b->n_gen_tiles > 0 || b->n_stream_zones > 0;
    }
    hipStream_t fs = b->stream;
    MarchGuard guard(b);
    if (fused) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        // A small batch runs its streamed remainder beside the fused launch (both are latency-bound and leave the chip
        // mostly empty); a large one runs them one after the other on the batch's stream: side by side the streamed
        // kernels wait for wavefront slots the fused launch holds for its whole duration.
        if (streamed && b->fused_stream != nullptr && b->n_surf <= 8192) {
            fs = b->fused_stream;
            HIP_TRY(hipEventRecord(b->ev_fork, b->stream));
            HIP_TRY(hipStreamWaitEvent(fs, b->ev_fork, 0));
            guard.forked = true;
        }
        if (b->timing) {
            e0 = next_event(b); e1 = next_event(b);
            if (!e0 || !e1) return fail(HEAT_E_DEVICE, "hipEventCreate failed");
            HIP_TRY(hipEventRecord(e0, fs));
        }
        rc = enqueue_fused(b, n_sub, fs, streamed && fs != b->stream);
        if (rc) return rc;
        if (b->timing) {
            HIP_TRY(hipEventRecord(e1, fs));
            b->ev_fused_pairs.push_back({e0, e1, n_sub});
        }
    }
    // (a sharded batch finishes the zones it owns only; beside a cluster-resident march only the zones no workgroup owns)
    const int zmode = fused ? 3 : (b->n_ranks > 1 ? 5 : 0);
    if (!streamed) {
        // nothing to stream
    } else if (b->timing && (b->timing_calls++ % b->timing_every) == 0) {
        for (int i = 0; i < n_sub; i++) {
            hipEvent_t e0 = next_event(b), e1 = next_event(b), e2 = next_event(b);
            if (!e0 || !e1 || !e2) return fail(HEAT_E_DEVICE, "hipEventCreate failed");
            HIP_TRY(hipEventRecord(e0, b->stream));
            enqueue_surfaces(b, -1, fused);
            HIP_TRY(hipEventRecord(e1, b->stream));
            enqueue_zones(b, zmode);
            HIP_TRY(hipEventRecord(e2, b->stream));
            b->ev_triples.push_back({e0, e1, e2});
        }
    } else if (b->use_graph) {
        // One graph holds a whole march call's sub-timesteps (up to 32; the usual call — ThermalModel::march runs a fixed
        // dt_subdivisions of them — replays it once): inside a graph consecutive kernels follow each other closer than
        // consecutive graph launches do. A call of another length re-captures; longer ones replay blocks.
        static const int per_graph_env = getenv("HEAT_AMD_GRAPH_SUBSTEPS") ? atoi(getenv("HEAT_AMD_GRAPH_SUBSTEPS")) : 0;
        // (a caller whose calls keep changing length gets the one-sub-timestep graph, which fits every length)
        const int want = std::max(1, b->graph_recaptures > 8 ? 1 : (per_graph_env > 0 ? std::min(per_graph_env, n_sub) : std::min(n_sub, 32)));
        if (!b->graph_exec || b->graph_fused != fused || b->graph_subs <= 0 || (b->graph_subs != want && n_sub % b->graph_subs != 0)) {
            if (b->graph_exec && b->graph_fused == fused) b->graph_recaptures++;
            // (a call of another length: the graph of the calls before may still be running — it goes only when it is done)
            if (b->graph_exec) HIP_TRY(hipStreamSynchronize(b->stream));
            if (b->graph_exec) { (void)hipGraphExecDestroy(b->graph_exec); b->graph_exec = nullptr; }
            if (b->graph) { (void)hipGraphDestroy(b->graph); b->graph = nullptr; }
            HIP_TRY(hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal));
            b->capturing = true;
            for (int i = 0; i < want; i++) {
                enqueue_surfaces(b, -1, fused);
                enqueue_zones(b, zmode);
            }
            b->capturing = false;
            HIP_TRY(hipStreamEndCapture(b->stream, &b->graph));
            HIP_TRY(hipGraphInstantiate(&b->graph_exec, b->graph, nullptr, nullptr, 0));
            b->graph_fused = fused;
            b->graph_subs = want;
        }
        int done = 0;
        for (; done + b->graph_subs <= n_sub; done += b->graph_subs) HIP_TRY(hipGraphLaunch(b->graph_exec, b->stream));
        for (; done < n_sub; done++) {  // (a remainder shorter than the graph: plain launches)
            enqueue_surfaces(b, -1, fused);
            enqueue_zones(b, zmode);
        }
    } else {
        for (int i = 0; i < n_sub; i++) {
            enqueue_surfaces(b, -1, fused);
            enqueue_zones(b, zmode);
        }
    }
    if (fused && fs != b->stream) {
        HIP_TRY(hipEventRecord(b->ev_fused, fs));
        HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_fused, 0));
        guard.joined = true;
    }
    HIP_TRY(hipGetLastError());
    return HEAT_OK;
}

int heat_batch_synchronize(heat_batch *b) {
    if (!b) return fail(HEAT_E_INVALID_ARG, "NULL batch");
    int rc = select_device(b);
    if (rc) return rc;
    // The failure flags travel to pinned memory at the end of the stream's work, so that ONE wait covers march and flags
    // (a synchronous copy behind the stream's wait is a second round trip to the device: 10-20 us of every short call).
    if (!b->h_flags) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_flags), 4 * sizeof(int)));
    HIP_TRY(hipMemcpyAsync(b->h_flags, b->d_flags.p, 4 * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    int f[4] = {b->h_flags[0], b->h_flags[1], b->h_flags[2], b->h_flags[3]};
    if (f[0]) {
        unsigned long long where;
        memcpy(&where, f + 2, sizeof where);
        HIP_TRY(hipMemset(b->d_flags.p, 0, 2 * sizeof(int)));
        HIP_TRY(hipMemset(b->d_flags.p + 2, 0xff, 2 * sizeof(int)));
        HIP_TRY(hipStreamSynchronize(nullptr));  // (the batch's streams do not wait for the null stream)
        // the first place (smallest number) and the kind that was seen there
        const int kinds = (int)(where & 0xff);
        const int64_t idx = (int64_t)(where >> 8);
        const bool zone = (kinds & FLAG_NAN_ZONE) && !(kinds & (FLAG_NAN_HS | FLAG_NAN_NOMASS | FLAG_UNREACHABLE));
        b->fail_kind = flags_to_status(kinds ? kinds : f[0]);
        b->fail_index = zone ? idx : ((idx >= 0 && idx < (int64_t)b->h_orig_of.size()) ? b->h_orig_of[idx] : -1);
        const int rc2 = flags_to_status(f[0]);
        const std::string what = heat::last_error();
        return fail(rc2, "%s; first seen at %s %lld", what.c_str(), zone ? "zone" : "surface", (long long)b->fail_index);
    }
    return HEAT_OK;
}

int heat_batch_failed_surface(const heat_batch *b, int64_t *index, int32_t *kind) {
    if (!b || !index || !kind) return fail(HEAT_E_INVALID_ARG, "NULL argument");
    *index = b->fail_index;
    *kind = b->fail_kind;
    return HEAT_OK;
}

int heat_batch_march(heat_batch *b, double *state, size_t n_state, const heat_weather *weather, int32_t n_sub,
                     const double *zone_a0, const double *zone_b0) {
    return heat_batch_march_ex(b, state, n_state, weather, n_sub, zone_a0, zone_b0, HEAT_OUT_ALL);
}

int heat_batch_march_ex(heat_batch *b, double *state, size_t n_state, const heat_weather *weather, int32_t n_sub,
                        const double *zone_a0, const double *zone_b0, int32_t what) {
    int rc = heat_batch_upload_inputs(b, state, n_state);
    if (rc) return rc;
    rc = heat_batch_march_resident(b, weather, n_sub, zone_a0, zone_b0);
    if (rc) return rc;
    rc = heat_batch_synchronize(b);
    if (rc) return rc;
    return download_impl(b, state, n_state, what, true);
}

int64_t heat_batch_nomass_iterations(heat_batch *b) {
    if (!b) return 0;
    if (hipSetDevice(b->device) != hipSuccess) return -1;
    if (hipStreamSynchronize(b->stream) != hipSuccess) return -1;
    std::vector<unsigned long long> h(b->d_nomass_iters.n);
    if (hipMemcpy(h.data(), b->d_nomass_iters.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) !=
        hipSuccess)
        return -1;
    unsigned long long v = 0;
    for (unsigned long long x : h) v += x;
    if (b->d_ucount.n) {
        h.resize(b->d_ucount.n);
        if (hipMemcpy(h.data(), b->d_ucount.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        for (unsigned long long x : h) v += x;
    }
    return (int64_t)v;
}

int heat_batch_set_timing(heat_batch *b, int32_t enabled) {
    if (!b) return fail(HEAT_E_INVALID_ARG, "NULL batch");
    b->timing = enabled != 0;
    b->timing_every = enabled > 1 ? enabled : 1;
    b->timing_calls = 0;
    b->ev_used = 0;
    b->ev_triples.clear();
    b->ev_fused_pairs.clear();
    return HEAT_OK;
}

int heat_batch_get_timing(heat_batch *b, double *surf_us, double *substep_us, int64_t *n_samples) {
    if (!b) return fail(HEAT_E_INVALID_ARG, "NULL batch");
    int rc = select_device(b);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    double s_surf = 0, s_all = 0, s_fused = 0;
    int64_t n_fused_steps = 0;
    const size_t n = b->ev_triples.size();
    for (const auto &t : b->ev_triples) {
        float ms1 = 0, ms2 = 0;
        HIP_TRY(hipEventElapsedTime(&ms1, t.e0, t.e1));
        HIP_TRY(hipEventElapsedTime(&ms2, t.e0, t.e2));
        s_surf += ms1;
        s_all += ms2;
    }
    for (const auto &p : b->ev_fused_pairs) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.e0, p.e1));
        s_fused += ms;
        n_fused_steps += p.n_sub;
    }
    const double stream_surf = n ? s_surf * 1000.0 / (double)n : 0.0;
    const double stream_all = n ? s_all * 1000.0 / (double)n : 0.0;
    const double fused_us = n_fused_steps ? s_fused * 1000.0 / (double)n_fused_steps : 0.0;
    // With a cluster-resident march the surface kernel of a sub-timestep is 1/n_sub of the fused launch (the
    // streamed remainder, if any, runs beside it on another stream).
    if (surf_us) *surf_us = n_fused_steps ? fused_us : stream_surf;
    // (a large batch runs its streamed remainder after the fused launch, a small one beside it)
    if (substep_us) *substep_us = n_fused_steps ? (b->n_surf > 8192 ? fused_us + stream_all : std::max(fused_us, stream_all)) : stream_all;
    if (n_samples) *n_samples = n_fused_steps ? n_fused_steps : (int64_t)n;
    b->ev_used = 0;
    b->ev_triples.clear();
    b->ev_fused_pairs.clear();
    return HEAT_OK;
}

int heat_batch_set_fusion(heat_batch *b, int32_t enabled) {
    if (!b) return fail(HEAT_E_INVALID_ARG, "NULL batch");
    b->fusion_on = enabled != 0;
    return HEAT_OK;
}

int64_t heat_batch_n_fused_surfaces(const heat_batch *b) { return b ? b->n_fused_surfaces : 0; }
int64_t heat_batch_n_fused_launches(const heat_batch *b) { return b ? b->n_fused_launches : 0; }

}  // extern "C"
