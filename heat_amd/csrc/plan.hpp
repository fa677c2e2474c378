// plan.hpp — host-only planning of a heat batch: descriptor checks, classification of the surfaces into kernel
// classes, zone-connected clusters and their workgroups (cluster-resident march), tiles, the packed constants in
// the device layout (layout.hpp), zone contribution lists, and the partition of a model over ranks.
//
// Nothing here touches a device: plan.cpp compiles with g++ as well as hipcc, so that the planner runs under
// AddressSanitizer / UBSan on the CPU (tests/test_planner_host.py) — heat_batch_create only uploads what
// make_plan produced.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/heat_amd.h"
#include "layout.hpp"

namespace heat {

constexpr int kMaxNodesGeneral = 4096;
constexpr int kScratchArrays = 7;
// fast classes: index = mi * 6 + nm * 3 + v;  M = 4 << mi;  nm: no-mass facings allowed;
// v = 0 per-node arrays, 1 palette constants, 2 palette + gas cavities between massive nodes
constexpr int kNumFast = 18;
extern const int kFastM[kNumFast], kFastNM[kNumFast], kFastPAL[kNumFast], kFastCAV[kNumFast];
constexpr int kSmall = kNumFast;         // all-no-mass surfaces of <= 4 nodes (general layout, register kernel)
constexpr int kSmallCav = kNumFast + 1;  // ... with a gas cavity (double glazing)
constexpr int kGeneral = kNumFast + 2;   // catch-all

// Everything heat_batch_create uploads, in device order.
struct Plan {
    int64_t n_surf = 0, n_zones = 0, n_state = 0, n_nodes = 0, n_cav = 0;
    double dt = 0;
    int64_t algorithmic_bytes = 0;
    int64_t class_counts[5] = {0, 0, 0, 0, 0};  // M4, M8, M16, small, general
    int64_t n_palette = 0;                      // surfaces whose constants are in palette form
    int64_t n_fused_surfaces = 0;

    // tiles
    std::vector<FastTile> fast_tiles[kNumFast];
    int n_stream_tiles[kNumFast] = {};  // tiles marched one sub-timestep per launch; the fused workgroups' tiles follow
    std::vector<GeneralTile> gen_tiles;  // [0, n_small_tiles) small (cavity-free first), the rest catch-all
    int n_small_tiles = 0, n_small_plain_tiles = 0, n_smallcav_stream_tiles = 0;
    int64_t gen_base = 0;       // first node slot of the general group
    int64_t node_slots = 0;     // total node slots incl. padding
    int64_t scratch_slots = 0;  // doubles of tri-diagonal scratch (catch-all kernel)

    // cluster-resident march: workgroup lists per class, index = width group (0: <= 4 wavefronts, 1: <= 8) + 2 * mixed
    std::vector<FusedBlock> fblocks[kNumFast][4];
    // teams of workgroups (clusters larger than one workgroup; layout.hpp, FusedSuper): the member blocks per class, the
    // clusters as runs of them, and per fused zone (fzones index) its exchange slot | members << 16
    std::vector<FusedBlock> team_blocks[kNumFast];
    std::vector<FusedSuper> team_supers[kNumFast];
    std::vector<uint32_t> team_zinfo;
    std::vector<int32_t> fzones, fzone_eoff;
    std::vector<uint16_t> fslots;
    std::vector<double> side_area;    // [2 * S]
    std::vector<int16_t> side_lzone;  // [2 * S]
    std::vector<int32_t> zone_block;  // zone -> fused workgroup (global number) or -1
    bool any_fused = false;

    // per-node constants (device layout)
    std::vector<double> V, U, alpha_f, alpha_b, mass, pal;
    std::vector<uint8_t> cls;
    int pal_stride = kPalNarrow;  // doubles per palette, and where its U entries start (layout.hpp)
    int pal_ubase = kPalVNarrow;
    std::vector<int32_t> cav_idx, cavref;
    std::vector<CavityDev> cavs;

    // per-surface records (device order)
    std::vector<int32_t> meta;
    std::vector<SideConst> side;
    std::vector<double> side_alpha, hs_fix;
    std::vector<int64_t> first_slot, slots;  // slots: 8 arrays of S
    std::vector<int64_t> dev_of, orig_of;    // original surface <-> device surface

    // zones
    std::vector<int64_t> zone_off, zone_slot;
    std::vector<double> zone_vol;
    std::vector<ZoneEntry> zone_entries;
    std::vector<uint8_t> touched;  // zones this batch's surfaces face

    // host copies used by download (original surface order)
    std::vector<int64_t> h_first_slot, h_node_count, h_out_slots[4];

    size_t nm_count_base[kNumFast + 1] = {};  // no-mass pass counters: one per tile of the NM fast classes, then
    size_t n_nm_counters = 0;                 // one per lane of the general-layout tiles
};

// Checks of heat_batch_create (reference: ThermalModel::new's Err / setup-time panics). Returns a heat_status.
int check_desc(const heat_batch_desc *d, std::string &err);
// The whole plan. Returns HEAT_OK or a negative heat_status with `err` set.
int make_plan(const heat_batch_desc *d, const heat_batch_options &opt, Plan &p, std::string &err);
// Internal consistency of a plan against its descriptor (every surface placed once, every index inside its
// array, every workgroup inside the kernel's limits). Used by the host-only tests; HEAT_OK or HEAT_E_SIZE.
int check_plan(const Plan &p, const heat_batch_desc *d, std::string &err);

// Zone-connected clusters (model.rs:556-590: surfaces exchange heat only through the zones they face): cluster id
// per surface (-1: faces no zone) and per zone, ids dense in [0, n_clusters).
void find_clusters(const heat_batch_desc *d, std::vector<int32_t> &cluster_of_surface,
                   std::vector<int32_t> &cluster_of_zone, int32_t &n_clusters);

// The message heat_last_error() returns (per thread).
std::string &last_error();

// heat_partition (include/heat_amd.h): rank of every surface, whole clusters kept together.
int partition_surfaces(const heat_batch_desc *d, int32_t n_ranks, int32_t *rank_of_surface, int64_t *n_shared_zones,
                       std::string &err);

// heat_batch_create_shard: the descriptor of the surfaces of one rank, with the arrays it points into.
struct ShardDesc {
    heat_batch_desc desc;
    std::vector<int64_t> original_index;  // surface q of the shard is surface original_index[q] of the model
    void build(const heat_batch_desc *d, const int32_t *rank_of_surface, int32_t rank);

  private:
    std::vector<int64_t> node_offset, i64[9];
    std::vector<double> mass, uvalue, front_alpha, back_alpha, f64[12];
    std::vector<int32_t> seg_cavity, i32[4];
};

}  // namespace heat
