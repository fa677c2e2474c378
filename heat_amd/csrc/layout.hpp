// layout.hpp — device data layout of a heat batch.
//
// All surfaces of a model are re-ordered into *groups* and *tiles*:
//
//   fast group (one per blocking factor M in {4, 8, 16})
//     tile  = the work of one 64-lane wavefront: G surfaces of k lanes each
//             (k = ceil(n / M) lanes per surface, G <= floor(64 / k));
//             lane l of the tile holds nodes [seg*M, seg*M + M) of surface l / k,
//             seg = l % k.
//     node arrays (T, V = dt/mass, U) of a tile are stored lane-blocked:
//             element (pair jp in [0, M/2), lane l) is a double2 at
//             node_base + (jp * Lk + l) * 2,   Lk = floor(64 / k) * k
//             so one wave instruction moves Lk * 16 contiguous bytes.
//   general group (catch-all: any chunk structure, cavities, per-node alphas)
//     tile  = 64 surfaces, one per lane; node j of lane l at node_base + j * 64 + l.
//
// Per-surface scalars are plain structure-of-arrays indexed by the device
// surface number d (tiles own contiguous ranges of d).
#pragma once
#include <stdint.h>

namespace heat {

constexpr int kWave = 64;
constexpr double kSigma = 5.670374419e-8;  // reference src/lib.rs:49
constexpr double kMassThreshold = 1e-5;    // reference src/discretization.rs:149,155

enum : int { KIND_SPACE = 0, KIND_AMBIENT = 1, KIND_OUTDOOR = 2 };

// Device-side numerical flags (OR-ed).
enum : int { FLAG_NAN_HS = 1, FLAG_NAN_NOMASS = 2, FLAG_NAN_ZONE = 4, FLAG_UNREACHABLE = 8,
              FLAG_EXCHANGE = 16 };  // a team of workgroups gave up waiting for a member's zone sums (kernels.hip)

struct FastTile {
    int64_t node_base;  // in doubles, into the unified node buffers
    int32_t surf_base;  // first device surface of the tile
    int16_t k;          // bits 0-7: lanes per surface; bit 8: every surface of the tile has n == k * M;
                        // bits 9-11: kind tags of the unified streamed list (kernels.hpp);
                        // bit 12 (kTileMixedBit): MIXED tile — bits 0-7 are the lanes the tile uses (Lk), its surfaces
                        // have lane counts of their own and follow each other in the lanes; lane l's surface and
                        // segment come from a table of 64 16-bit entries behind the tile's class bytes, at
                        // cls + node_base + M * Lk:  bits 0-5 surface of the tile, 6-11 segment, 12 last segment.
                        // (The workgroups of the cluster-resident march pack their clusters' walls this way.)
    int16_t G;          // surfaces in this tile
};
constexpr int kTileMixedBit = 1 << 12;
constexpr int kTileChunkyBit = 1 << 13;  // FastTile::k: the tile holds no-mass chunks other than one-node facings (kernels.hip)
constexpr int kLaneLastBit = 1 << 12;
constexpr int kLaneTableSlots = 128;  // node slots a mixed tile takes behind its M * Lk for the table (128 bytes used)

struct GeneralTile {
    int64_t node_base;  // in doubles
    int32_t surf_base;
    int32_t G;          // surfaces in this tile (<= 64)
    int32_t n_max;      // max node count in the tile
    int32_t pad;
    int64_t scratch_base;  // in doubles, into the scratch buffer; [array][n_max][64]
};

// Per-surface data is kept per SIDE (front = side 0, back = side 1), record index = side * S + d
// for device surface d, so that the lane that evaluates a side fetches it with a few 16-byte loads.
//
// Constants of one side (64 bytes).
struct alignas(16) SideConst {
    int32_t kind_n;   // bits 0-1: boundary kind; bit 2: always windward (|cos tilt| >= 0.98,
                      // reference src/surface.rs:38); bits 4-5: kind of the surface's OTHER side; bits 16-31: node
                      // count of the surface
    int32_t zone;     // zone index when kind == KIND_SPACE; always a valid index (0 otherwise: the kernels gather the
                      // zone temperature unconditionally). A BACK side of kind KIND_AMBIENT carries the front side's
                      // boundary source here (zone) and in `forced` (ambient temperature): surface.rs:672-686 takes
                      // t_front for its radiant temperature
    double ambient;   // kind == KIND_AMBIENT: Boundary::AmbientTemperature { temperature };
                      // kind == KIND_SPACE: the bits of an int64 — this side's position in the zone contribution
                      // list (ZoneEntry / SideArrays::zc), see side_entry_pos()
    double emis;      // thermal emissivity of this face (src/surface.rs:335,338)
    double alpha;     // general / small classes: solar absorptance of the face node (front_alphas[0] / back_alphas[n-1]).
                      // FAST classes: TARP natural-convection coefficient for air COLDER than the surface, h / |dT|^(1/3)
                      // (the absorptance is folded into SideDyn::solar at upload)
    double cos_eff;   // general / small classes: cos_surface_tilt as this side's ConvectionParams takes it (front
                      // Outdoor: -cos). FAST classes: the coefficient for air WARMER than the surface. Both are
                      // 1.31, 9.482 / (7.238 - |cos|) or 1.81 / (1.382 + |cos|) by the branches of convection.rs:87-110.
    double forced;    // kind == KIND_OUTDOOR: 2.537 * 1.67 * sqrt(perimeter * wind_modifier / area)  (src/convection.rs:157-163);
                      // kind == KIND_SPACE: the surface's area (model.rs:562-585: what the side's coefficient is weighted with)
    double nx, ny;    // surface normal (for is_windward)
};
// Inputs other modules write between marches, converted once at upload.
struct alignas(16) SideDyn {
    double solar;     // incident solar irradiance, clamped as src/surface.rs:916-923 does; FAST classes: times the
                      // face absorptance (the only node that absorbs on the fast path)
    double rad_t;     // (ir / sigma)^0.25 - 273.15 (src/surface.rs:647,692)
};
// Outputs of iterate_surfaces (src/model.rs:154-169).
struct alignas(16) SideOut {
    double hs;        // convection coefficient
    double flow;      // convective heat flow
};

// A zone-facing side's share of calculate_zones_abc (model.rs:562-585): the new convection coefficient times the surface's
// area (a Space-facing side carries its area in SideConst::forced) and the new face temperature — k_zones sums these 16
// bytes per entry and reads nothing else.
struct alignas(16) ZoneContrib {
    double ha;
    double t_face;
};

struct SideArrays {
    const SideConst *sc;   // [2 * S]
    const SideDyn *dyn;    // [2 * S]
    SideOut *out;          // [2 * S]
    const double *hs_fix;  // [2 * S] debug overrides (src/surface.rs:374-380), NaN = none; nullable
    ZoneContrib *zc;       // [zone entries]: what each zone-facing side adds to its zone's heat balance, written by the
                           // surface kernels at the side's position in the zone's list (model.rs:562-585)
    int32_t S;
    int32_t pad;
};

// Unified per-node buffers (all groups).
// Palette form of the per-node constants (fast classes with PAL = 1): a wall has only a handful of
// distinct V = dt/mass and U values (per layer one of each, plus a V per joint and end node), so each node
// stores a one-byte class and each surface a palette of V-values followed by U-values (entry 0 of both is 0.0:
// padding, no-mass, Back). The class byte:
//   bits 0-3  index of V; 14 and 15 mark the first node of a no-mass chunk of one / of two nodes (V = 0)
//   bits 4-6  index of U
// The batch's palettes are stored in the width its walls need (NodeArrays::pal_stride, pal_ubase):
//   tiny    4 V + 4 U =  8 doubles (64 bytes per surface): walls of one or two materials (three distinct V and U besides
//          the 0.0 of entry 0) — BASELINE config 3 and the uniform headline; 32 bytes per surface less to stream
//   narrow  8 V + 4 U = 12 doubles (96 bytes per surface): walls of up to three layers
//   wide   16 V + 8 U = 24 doubles (192 bytes per surface): up to 13 distinct V and 7 distinct U — walls of six
//          different layers (render / brick / insulation / block / service gap / plaster) stay in palette form, and
//          with it candidates for the cluster-resident march
constexpr int kPalV = 16;            // wide (what index fields and limits are sized for)
constexpr int kPalU = 8;
constexpr int kPal = kPalV + kPalU;
constexpr int kPalVNarrow = 8;
constexpr int kPalUNarrow = 4;
constexpr int kPalNarrow = kPalVNarrow + kPalUNarrow;
constexpr int kPalVTiny = 4;
constexpr int kPalUTiny = 4;
constexpr int kPalTiny = kPalVTiny + kPalUTiny;
constexpr int kPalVMark1 = 14;       // V index values that are chunk marks; V entries usable: 0 .. 13
constexpr int kPalUShift = 4;

struct NodeArrays {
    double *T;               // node temperatures (state)
    const double *V;         // dt / mass for massive nodes, 0 for no-mass and padding
    const double *U;         // Solid u, 0 for Back/padding (cavity segments: 0, see cav)
    const uint8_t *cls;      // PAL classes: class byte of node (lane l, j) at node_base + l * M + j
    const double *pal;       // PAL classes: palette of device surface d at pal + d * pal_stride
    const int32_t *cavref;   // CAV classes: per device surface {node, cavity, node, cavity}: up to two gas
                             // cavities between massive nodes (segment node -> node+1), -1 = none
    const struct CavityDev *cavs;
    const double *alpha_f;   // general group only (same indexing), else nullptr
    const double *alpha_b;
    const int32_t *cav;      // general group only: cavity index or -1
    const double *mass;      // general group only: raw thermal mass
    int32_t pal_stride;      // doubles per palette: kPalNarrow or kPal
    int32_t pal_ubase;       // where its U entries start: kPalVNarrow or kPalV
};

struct CavityDev {
    double thickness, height, angle, eout, ein;
    int32_t gas, pad;
};

// Weather and per-march constants, device-resident. step is advanced by the zone kernel.
struct StepWeather {
    double t_out;    // dry bulb
    double sqrt_ws;  // sqrt(wind speed): the forced coefficient is wf * SideConst::forced * sqrt_ws
    double sin_wd, cos_wd;
};

// Cluster-resident march ("fused"): one workgroup owns whole zone-connected clusters — every surface
// that faces one of its zones — so the zone balance of a sub-timestep is done in LDS and the node temperatures
// stay in registers over all sub-timesteps of a march.
constexpr int kFusedMaxWaves = 8;    // tiles (wavefronts) per workgroup
constexpr int kFusedMaxZones = 32;   // zones per workgroup
constexpr int kFusedMaxEntries = 1024;  // zone-facing sides per workgroup
struct FusedBlock {
    int32_t first_tile;  // into the class's tile list; the block's tiles are contiguous
    int32_t n_tiles;     // <= kFusedMaxWaves
    int32_t first_zone;  // into FusedArgs::zones / zone_eoff
    int32_t n_zones;     // <= kFusedMaxZones
    int32_t first_small; // small all-no-mass surfaces of the block's clusters (glazing, thin walls): tiles of the general
    int32_t n_small;     // layout, one wavefront each, after the fast-path wavefronts; n_tiles + n_small <= kFusedMaxWaves
};
// A contribution to a zone's heat balance (model.rs:562-585) is the LDS slot of the side that makes it:
//   slot = side * (64 * W) + wave_in_block * 64 + lane of the side's owner   (W = 4 or 8, the block's width group)
// listed per zone in the reference's order.
struct FusedArgs {
    const FusedBlock *blocks;
    const int32_t *zones;       // global zone number of the block-local zone (first_zone + j)
    const int32_t *zone_eoff;   // slots of that zone: [zone_eoff[first_zone + j], zone_eoff[first_zone + j + 1])
    const uint16_t *slots;
    const double *side_area;    // [2 * S]: area of the surface, per side record
    const int16_t *side_lzone;  // [2 * S]: block-local zone of a side, -1 unless the side faces a Space
    const double *a0, *b0, *vol;
    double *zone_T;
    double dt;
    int32_t n_sub;              // sub-timesteps marched by one launch
    int32_t pad;
    // Work queue (sharded batches): the launch holds fewer workgroups than blocks; a workgroup that has finished
    // its cluster set takes the next one: gridDim.x + atomicAdd(queue, 1). nullptr: one workgroup per FusedBlock.
    unsigned int *queue;
    int32_t n_blocks;
    int32_t pad2;
    const GeneralTile *gen_tiles;        // workgroups with small surfaces
    int64_t gen_base;
    unsigned long long *small_iters;     // no-mass pass counters of the general-layout tiles ([tile][lane])
    // teams (clusters larger than a workgroup; below)
    const struct FusedSuper *supers;
    const uint32_t *team_zinfo;
    unsigned long long *xbuf;            // the teams' exchange areas
    int32_t n_super;
    uint32_t tag_base;                   // launch number << 22 (tag = tag_base | round << 12 | sub-timestep + 1)
    int32_t team_size;                   // workgroups per team in this launch: the most members a cluster of the list has
    int32_t reverse;     // walk the tile list (streamed launches) / the block or cluster list (cluster-resident) from its end: zig-zag sweeps, batch.hip
};

// Clusters larger than a workgroup (a building whose rooms are all joined by interior walls): a TEAM of up to
// kTeamMax workgroups of four wavefronts marches the cluster together. Every member holds part of the walls and the
// zones they face; a zone faced from several members is balanced from all their partial sums, which the members
// exchange through L2 once per sub-timestep as 8-byte {32 bits of data, 32-bit tag} granules (one `sc1` store each;
// the tag names launch, round and sub-timestep, so a granule is complete when its tag is) — no flag, no fence.
//   team_zinfo[first_zone + j]  low 16 bits: the zone's slot in the team's exchange area (its number in the cluster);
//                               high bits: which members face it (the only ones whose sums are awaited)
//   exchange area of a team     [parity of the sub-timestep][kTeamZones][kTeamMax][4 granules: a lo, a hi, b lo, b hi]
// Co-residency (a member must not wait for a workgroup that cannot start): the launch holds exactly n_teams x team_size
// workgroups (team_size: the most members a cluster of the launch has), never more than the chip takes at once (checked on the host), and every team walks the clusters
// team, team + n_teams, ... in lockstep. Every wait is bounded: on expiry FLAG_EXCHANGE is raised and the march goes
// on unsynchronised — the host reports HEAT_E_DEVICE.
constexpr int kTeamMax = 8;      // workgroups per team
constexpr int kTeamZones = 256;  // zones per cluster marched by a team
struct FusedSuper {
    int32_t first_block;  // into the team list of FusedBlocks; the members' blocks are contiguous
    int32_t n_members;    // <= kTeamMax
};

struct ZoneEntry {
    uint32_t t_index;   // index into NodeArrays::T of the face node
    uint32_t hs_index;  // side record index (side * S + d) into SideArrays::out
    double area;
};
// (k_zones reads none of it: coefficient x area and face temperature of entry e come from SideArrays::zc[e], which the
// surface kernels fill — contiguous per zone instead of two gathers per entry. The entries stay as the host's record of
// who contributes where.)

}  // namespace heat
