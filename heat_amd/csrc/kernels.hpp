// kernels.hpp — host-callable launchers of the kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.hpp"

namespace heat {

// SimulationState slot numbers per device surface (used only by upload/download).
struct SlotArrays {
    const int64_t *hs_f, *hs_b, *flow_f, *flow_b, *solar_f, *solar_b, *ir_f, *ir_b;
};

// nm: 0 all-massive, 1 one-node no-mass facings, 2 any no-mass chunk of one or two nodes (palette classes)
void launch_surfaces_fast(int M, int nm, int pal, int cav, double grid_share, const FastTile *tiles, int n_tiles,
                          const NodeArrays &na,
                          const SideArrays &sa, const StepWeather *weather, const int *step_ptr, int step_fixed,
                          const double *zone_T, int *flags, unsigned long long *nomass_iters, int n_cu, hipStream_t st, int reverse = 0);
// Tile kinds of the unified streamed list (FastTile::k, bits 9-11; k_surfaces_stream)
constexpr int kStreamKindShift = 9;
constexpr int kStreamKindSmall = 3;
constexpr int kStreamNmBit = 1 << 11;
// variant: 0 tiles of 16 nodes per lane, 1 tiles of 8 / 4 nodes per lane with one-node facings at most + small surfaces,
// 2 tiles of 8 / 4 nodes per lane with other no-mass chunks, 3 tiles with gas cavities — double glazing and walls of
// 8 / 4 nodes per lane (kernels.hip, k_surfaces_stream)
constexpr int kStreamVariants = 4;
void launch_surfaces_stream(int variant, const FastTile *tiles, int n_tiles, const NodeArrays &na, int64_t gen_base, const SideArrays &sa,
                            const StepWeather *weather, const int *step_ptr, int step_fixed, const double *zone_T,
                            int *flags, unsigned long long *nomass_iters, int n_cu, hipStream_t st, int reverse = 0);
int fused_blocks_per_cu(int M, int cav, int mixed, int max_waves, int pal_stride);
// ... of a team variant, from the runtime's occupancy query for that very kernel (0: the query failed)
int fused_team_blocks_per_cu(int M, int nm, int pal_stride);
hipError_t launch_surfaces_fused(int M, int nm, int cav, int mixed, int max_waves, int grid_blocks, const FastTile *tiles, int n_tiles,
                                 const NodeArrays &na, const SideArrays &sa, const StepWeather *weather, int *flags,
                                 unsigned long long *nomass_iters, const FusedArgs &fa, hipStream_t st);
void launch_surfaces_general(const GeneralTile *tiles, int n_tiles, const NodeArrays &na, int64_t gen_base,
                             const SideArrays &sa, const CavityDev *cavs, double *scratch,
                             const StepWeather *weather, const int *step_ptr, int step_fixed,
                             const double *zone_T, int *flags, unsigned long long *nomass_iters, hipStream_t st);
void launch_surfaces_small(int with_cavities, const GeneralTile *tiles, int n_tiles, const NodeArrays &na,
                           int64_t gen_base, const SideArrays &sa, const CavityDev *cavs,
                           const StepWeather *weather, const int *step_ptr, int step_fixed, const double *zone_T,
                           int *flags, unsigned long long *nomass_iters, hipStream_t st);
void launch_zones(const int64_t *zone_off, const ZoneEntry *entries, const ZoneContrib *zc,
                  const double *a0, const double *b0, const double *zone_vol, double *zone_T, double *partial,
                  int n_zones, double dt, int *step_ptr, int *flags, int mode, const int32_t *zlist, int n_list,
                  const int32_t *slot_of, int n_shared, int rows, hipStream_t st);
void launch_zone_update_shared(const double *gathered, int n_blocks, const int32_t *shared_zone, int n_shared,
                               const double *a0, const double *b0, const double *zone_vol, double *zone_T,
                               double dt, int *flags, hipStream_t st);
void launch_zone_update(const double *gathered, int n_blocks, const double *a0, const double *b0,
                        const double *zone_vol, double *zone_T, int n_zones, double dt, int *step_ptr,
                        int *flags, hipStream_t st);
void launch_nodes_fast(int M, const FastTile *tiles, int n_tiles, double *Tbuf, const int32_t *meta,
                       const int64_t *first_slot, double *state, int to_state, const uint8_t *cls, hipStream_t st);
void launch_nodes_general(const GeneralTile *tiles, int n_tiles, double *Tbuf, const int32_t *meta,
                          const int64_t *first_slot, double *state, int to_state, hipStream_t st);
void launch_surf_scalars(int n_surf, const SlotArrays &sl, SideDyn *dyn, SideOut *out, const double *side_alpha,
                         double *state, int to_state, int what, hipStream_t st);
void launch_zone_scalars(int n_zones, const int64_t *zone_slot, double *zone_T, double *state, int to_state,
                         hipStream_t st);
void launch_inputs_compact(int n_surf, int n_zones, const double *in, const double *side_alpha, SideDyn *dyn, double *zone_T,
                           const SlotArrays &sl, double *mirror, hipStream_t st);
void launch_outputs_compact(int n_surf, int n_zones, const SideOut *out, const int32_t *orig_of, const double *zone_T,
                            double *dst, hipStream_t st);
void launch_begin_march(const StepWeather *h_weather, StepWeather *weather, int n_sub, const double *h_zone_ab, double *a0,
                        double *b0, int n_zones, int *step_ptr, hipStream_t st);
void launch_set_step(int *step_ptr, int v, int last, hipStream_t st);

}  // namespace heat
