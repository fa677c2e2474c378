// kernels.hip — gfx950 kernels of the wall heat-conduction path.
//
//   k_surfaces_fast<M>   iterate_surfaces (reference src/model.rs:102-180) for all-massive,
//                        solid-conductance, opaque surfaces: lane-blocked RK4 stencil in registers,
//                        M nodes per lane, neighbour exchange by wave shuffles.
//   k_surfaces_general   the same reference function for ANY surface (no-mass chunks, gas
//                        cavities, per-node solar absorption): one lane per surface, tri-diagonal
//                        matrices in a global scratch area, operation order of the reference.
//   k_zones              calculate_zones_abc + estimate_zones_future_temperatures
//                        (src/model.rs:489-597,650-674): one wavefront per zone.
//   k_zone_update        the zone update from gathered per-rank partial sums (multi-GPU).
//   k_gather_* / k_scatter_*   SurfaceTrait slot accessors (src/surface_trait.rs:81-164) in bulk.
//
// No MFMA anywhere: this is an HBM-bound f64 stencil.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "device_math.hpp"
#include "kernels.hpp"
#include "layout.hpp"

namespace heat {

// ---------------------------------------------------------------------------
// Boundary conditions of one side — reference src/surface.rs:596-717.
//   c         the side's constants (kind, effective cos tilt, forced coefficient, normal)
//   air_t     boundary temperature of this side (get_boundary_temperature, model.rs:79-96)
//   rad_alt   rad_temperature for non-Outdoor kinds (t_front / t_back / quirk, surface.rs:616,631,665,676)
//   rad_out   rad_temperature for Outdoor: (ir/sigma)^0.25 - 273.15, precomputed at upload
//   surf_t    surface temperature the reference reads from `state` for this side
__device__ __forceinline__ void eval_side(const SideConst &c, const StepWeather &w, double air_t, double rad_alt,
                                          double rad_out, double surf_t, double &hs, double &rad_t, int &bad) {
    const double natural = tarp_natural(air_t, surf_t, c.cos_eff, bad);
    if ((c.kind_n & 3) == KIND_OUTDOOR) {
        // is_windward, surface.rs:37-46
        const bool windward = (c.kind_n & 4) ? true : ((c.nx * w.sin_wd + c.ny * w.cos_wd) > 0.0);
        const double wf = windward ? 1.0 : 0.5;
        hs = wf * (c.forced * w.sqrt_ws) + natural;  // convection.rs:161-167
        rad_t = rad_out;
    } else {
        hs = natural;
        rad_t = rad_alt;
    }
}

__device__ __forceinline__ double boundary_temperature(const SideConst &c, const StepWeather &w,
                                                       const double *__restrict__ zone_T) {
    const int kind = c.kind_n & 3;  // get_boundary_temperature, model.rs:79-96
    if (kind == KIND_SPACE) return zone_T[c.zone];
    if (kind == KIND_AMBIENT) return c.ambient;
    return w.t_out;
}

// A Space-facing side's position in its zone's contribution list (stored in the `ambient` slot, layout.hpp).
__device__ __forceinline__ long long side_entry_pos(const SideConst &c) { return __double_as_longlong(c.ambient); }
__device__ __forceinline__ void put_zone_contrib(const SideArrays &sd, const SideConst &c, double hs, double t_face) {
    if ((c.kind_n & 3) == KIND_SPACE) {
        ZoneContrib z;
        z.ha = hs * c.forced;  // (a Space-facing side's `forced` slot holds the area, layout.hpp)
        z.t_face = t_face;
        sd.zc[side_entry_pos(c)] = z;
    }
}

// A numerical failure (the reference panics there: surface.rs:704-707,850; model.rs:417-420): the kind goes into
// the flag word, and the FIRST place it happened — smallest device surface (or zone) number — into the 64-bit word
// behind it: number << 8 | kind bits. Rare by nature: no cost on the good path.
__device__ __forceinline__ void report_failure(int *flags, int bad, unsigned int index) {
    atomicOr(flags, bad);
    atomicMin(reinterpret_cast<unsigned long long *>(flags + 2), ((unsigned long long)index << 8) | (unsigned int)(bad & 0xff));
}

__device__ __forceinline__ double shfl_f64(double v, int src_lane) { return __shfl(v, src_lane, kWave); }

// Neighbour exchange across the whole wavefront as DPP rotates (VALU speed; a ds_bpermute round trip through the
// LDS crossbar costs ~100 cycles of dependent latency per RK stage). from_prev: lane l receives lane l - 1's
// value (lane 0 receives lane 63's); from_next: lane l receives lane l + 1's (lane 63 receives lane 0's).
template <int CTRL>
__device__ __forceinline__ double dpp_rotate_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// Sum over the wavefront with DPP (row shifts inside the rows of 16 lanes, then the two row broadcasts): the
// total arrives in lane 63 and is handed to every lane. A fixed tree: run-to-run deterministic, the same in
// k_zones and in the cluster-resident march.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v = dpp_add_f64<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add_f64<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add_f64<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add_f64<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_add_f64<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3 add the sum of the row before
    v = dpp_add_f64<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3 add the sum of rows 0-1
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Sums inside the rows of 16 lanes: lane 15 of every row holds its row's sum (the first four steps of wave_sum_f64).
__device__ __forceinline__ double row_sum_f64(double v) {
    v = dpp_add_f64<0x111, 0xf>(v);
    v = dpp_add_f64<0x112, 0xf>(v);
    v = dpp_add_f64<0x114, 0xf>(v);
    v = dpp_add_f64<0x118, 0xf>(v);
    return v;
}

__device__ __forceinline__ double from_prev_lane(double v) { return dpp_rotate_f64<0x13C>(v); }  // wave_ror:1
__device__ __forceinline__ double from_next_lane(double v) { return dpp_rotate_f64<0x134>(v); }  // wave_rol:1

// ---------------------------------------------------------------------------
// Fast path. One wavefront per tile; see layout.hpp for the lane blocking.
//
// RK4 (surface.rs:228-308) on  dT/dt = A T + q,  (A x)_j = V_j (f_j - f_{j-1}),  f_j = U_j (x_{j+1} - x_j)  — the
// tri-diagonal system of get_k_q + rearrange_k (discretization.rs:596-700, surface.rs:168-187) in flux form. The two
// faces enter as fluxes too: the front face is the "flux from the left" of node 0, hF x_0 - qF, the back face the
// "flux to the right" of the last node, qB - hB x_last (the K[0,0] -= hs / q[0] += ... terms of get_k_q,
// discretization.rs:658-697, before the row scaling by dt/C).
//
// K' and q' are FROZEN over the four stages (surface.rs:268-293), so the stages telescope: with w = A T + q (= k1)
//   k2 = w + A w / 2,  k3 = w + A k2 / 2,  k4 = w + A k3     (surface.rs:280-292)
//   T + (k1 + 2 k2 + 2 k3 + k4) / 6 = T + w + A w / 2 + A^2 w / 6 + A^3 w / 24        (surface.rs:296-305)
// and, in Horner form with v = w / 24:
//   z2 = 4 v + A v,   z3 = 12 v + A z2,   T_new = T + 24 v + A z3.
// Four applications of A as before, but no per-stage accumulator: 20 f64 operations per node and sub-timestep instead
// of 24, the same values up to rounding (1e-15 relative against the oracle's literal stages; tested at 1e-9).
// Every application consumes its input in place: x_j is dead once f_j is formed.
//
// The faces cost no select inside the stages: the caller hands in EFFECTIVE conductances —
//   Ue[last node] = hB (the reference's UValue::Back there is 0) with the node "behind" it held at 0: padding slots
//                   are 0 in T, v and z (V = 0 keeps them there), and a surface's last lane takes 0 for its right neighbour;
//   ULe = hF for a surface's first lane, which takes 0 for its left neighbour —
// so that f_last = hB (0 - x_last) and f_{-1} = hF (x_0 - 0) fall out of the plain stencil, wherever the last node
// sits inside its lane. The sources qF, qB enter the first application only (AFFINE), qB through qsel(j).
// No value of a neighbouring surface (possibly NaN) ever enters: the zeros above replace exactly those.
//   put(j, y): receives (A x [+ q])_j.
template <int M, bool AFFINE, typename VF, typename QF, typename PUT>
__device__ __forceinline__ void apply_A(const double (&x)[M], VF V, const double (&Ue)[M], double ULe, bool is_first,
                                        bool is_last, double qF, QF qsel, PUT put) {
    const double xl_raw = from_prev_lane(x[M - 1]);
    const double xr_raw = from_next_lane(x[0]);
    const double xl = is_first ? 0.0 : xl_raw;
    const double xr = is_last ? 0.0 : xr_raw;
    double fprev = ULe * (x[0] - xl);
    if (AFFINE) fprev -= qF;
#pragma unroll
    for (int j = 0; j < M; j++) {
        const double xj = x[j];
        double f = Ue[j] * (((j == M - 1) ? xr : x[j + 1]) - xj);
        if (AFFINE) f += qsel(j);
        put(j, V(j) * (f - fprev));
        fprev = f;
    }
}

// qF: 0 unless this lane is its surface's first; qsel(j): qB at the surface's last node, else 0.
template <int M, typename VF, typename QF>
__device__ __forceinline__ void rk4_horner(double (&T)[M], VF V, const double (&Ue)[M], double ULe, bool is_first,
                                           bool is_last, double qF, QF qsel) {
    double v[M], z[M];
    apply_A<M, true>(T, V, Ue, ULe, is_first, is_last, qF, qsel, [&](int j, double y) { v[j] = y * (1.0 / 24.0); });
    apply_A<M, false>(v, V, Ue, ULe, is_first, is_last, qF, qsel, [&](int j, double y) { z[j] = 4.0 * v[j] + y; });
    apply_A<M, false>(z, V, Ue, ULe, is_first, is_last, qF, qsel, [&](int j, double y) { z[j] = 12.0 * v[j] + y; });
    apply_A<M, false>(z, V, Ue, ULe, is_first, is_last, qF, qsel,
                      [&](int j, double y) { T[j] = (T[j] + 24.0 * v[j]) + y; });
}

// ---------------------------------------------------------------------------
// Small all-no-mass surfaces (n <= 4: single-layer no-mass walls, double glazing with its gas
// cavity): one lane per surface, the whole chunk (0, n) in registers. Same layout as the general
// group, reference operation order (march_nomass, surface.rs:790-898), no FMA contraction.
#pragma clang fp contract(off)
constexpr int kSmallNodes = 4;
// The node temperatures of a small surface as four scalars. (As `double T[4]` the array stays in memory — LDS, or
// scratch inside k_surfaces_stream: the compiler turns the select chain that picks the last node into an indexed load
// — and every pass of the no-mass loop then waits for a memory round trip.) j is a constant after unrolling.
__device__ __forceinline__ double opaque_select(bool cond, double x, double y) {  // cond ? x : y
    const unsigned long long m = __builtin_amdgcn_ballot_w64(cond);
    int lo, hi;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(y)), "v"(__double2loint(x)), "s"(m));
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(y)), "v"(__double2hiint(x)), "s"(m));
    return __hiloint2double(hi, lo);
}
struct Nodes4 {
    double a, b, c, d;
    __device__ __forceinline__ double get(int j) const { return j == 0 ? +a : (j == 1 ? +b : (j == 2 ? +c : +d)); }
    __device__ __forceinline__ void set(int j, double v) {
        if (j == 0) a = v;
        else if (j == 1) b = v;
        else if (j == 2) c = v;
        else d = v;
    }
    __device__ __forceinline__ double last(int nn) const {  // node nn - 1
        // (selects the optimizer cannot see through: it turns a plain select chain over a, b, c, d into a load indexed
        // by nn - 1, which keeps the four values in memory)
        double r = a;
        r = opaque_select(nn == 2, b, r);
        r = opaque_select(nn == 3, c, r);
        r = opaque_select(nn == 4, d, r);
        return r;
    }
};

// One sub-timestep of one small surface: calc_border_conditions, march_nomass on the chunk (0, nn), the
// coefficients and flows with the new temperatures (model.rs:150-169). T is updated in place.
template <int CAV>
__device__ __forceinline__ void small_step(const SideConst &cf, const SideConst &cb, const SideDyn &df, const SideDyn &db,
                                           const double *__restrict__ hs_fix, int d, int S, const StepWeather &w,
                                           double t_front_b, double t_back_b, int nn, Nodes4 &T,
                                           const double (&Us)[kSmallNodes], const double (&sol)[kSmallNodes],
                                           const int (&cav)[kSmallNodes], const CavityDev (&cv)[kSmallNodes - 1], int &bad,
                                           unsigned int &iters, SideOut &of, SideOut &ob) {
    constexpr int NS = kSmallNodes;
    const int bk = cb.kind_n & 3;
    auto last = [&](const double (&x)[NS]) {
        double r = x[0];
#pragma unroll
        for (int j = 1; j < NS; j++) r = (j == nn - 1) ? +x[j] : +r;
        return r;
    };
    const double T0 = T.a, Tn = T.last(nn);
    const bool quirk = (bk == KIND_AMBIENT);
    double f_hs, f_rad, b_hs, b_rad;
    const double f_surf = T0, b_surf = quirk ? T0 : Tn;
    eval_side(cf, w, t_front_b, t_front_b, df.rad_t, f_surf, f_hs, f_rad, bad);
    eval_side(cb, w, t_back_b, quirk ? t_front_b : t_back_b, db.rad_t, b_surf, b_hs, b_rad, bad);
    if (f_hs != f_hs || b_hs != b_hs) bad |= FLAG_NAN_HS;
    if (hs_fix != nullptr) {
        const double ff = hs_fix[d], fb = hs_fix[S + d];
        if (ff == ff) f_hs = ff;
        if (fb == fb) b_hs = fb;
    }
    const double f_radhs = rad_hs(cf.emis, f_rad, f_surf);
    const double b_radhs = rad_hs(cb.emis, b_rad, b_surf);

    double old_err = 99999.;
    int count = 0;
    for (;;) {
        // get_k_q for the chunk (0, nn) — discretization.rs:596-700
        double lo[NS], dg[NS], up[NS], q[NS];
#pragma unroll
        for (int j = 0; j < NS; j++) { lo[j] = 0.0; dg[j] = 0.0; up[j] = 0.0; q[j] = 0.0; }
#pragma unroll
        for (int j = 0; j < NS - 1; j++) {
            if (j < nn - 1) {
                double u = Us[j];
                if constexpr (CAV) {
                    if (cav[j] >= 0) u = cavity_u_value(cv[j], T.get(j), T.get(j + 1), bad);
                }
                dg[j] += -u;
                dg[j + 1] = dg[j + 1] - u;
                up[j] = up[j] + u;
                lo[j + 1] = lo[j + 1] + u;
            }
        }
        q[0] += t_front_b * f_hs + f_radhs * (f_rad - T.a);
        dg[0] += -f_hs;
        const double bq = t_back_b * b_hs + b_radhs * (b_rad - T.last(nn));
#pragma unroll
        for (int j = 0; j < NS; j++) {
            if (j == nn - 1) { q[j] += bq; dg[j] += -b_hs; }
        }
        iters++;
#pragma unroll
        for (int j = 0; j < NS; j++) q[j] = (q[j] + sol[j]) * -1.;  // surface.rs:828-832
        // mut_n_diag_gaussian(q, 3)
#pragma unroll
        for (int j = 1; j < NS; j++) {
            if (j < nn) {
                const double f = lo[j] / dg[j - 1];
                dg[j] -= f * up[j - 1];
                q[j] -= f * q[j - 1];
            }
        }
        double x[NS];
#pragma unroll
        for (int j = NS - 1; j >= 0; j--) {
            if (j == nn - 1) x[j] = q[j] / dg[j];
            else if (j < nn - 1) x[j] = (q[j] - up[j] * x[(j + 1) % NS]) / dg[j];
            else x[j] = 0.0;
        }
        double err = 0.0;
#pragma unroll
        for (int j = 0; j < NS; j++) if (j < nn) err += fabs(x[j] - T.get(j));
        if (err > old_err) break;                            // surface.rs:842-848
        if (err != err) { bad |= FLAG_NAN_NOMASS; break; }   // surface.rs:850
#pragma unroll
        for (int j = 0; j < NS; j++) if (j < nn) T.set(j, (T.get(j) + x[j]) * 0.5);
        const double tol = (count < 100) ? 0.01 : 0.5;      // surface.rs:885
        if (err / (double)nn < tol) break;
        old_err = err;
        count++;
    }

    {   // outputs with the new surface temperatures (model.rs:150-169)
        const double T0n = T.a, Tnn = T.last(nn);
        double fh, bh, r_;
        eval_side(cf, w, t_front_b, t_front_b, df.rad_t, T0n, fh, r_, bad);
        eval_side(cb, w, t_back_b, t_back_b, db.rad_t, quirk ? T0n : Tnn, bh, r_, bad);
        if (fh != fh || bh != bh) bad |= FLAG_NAN_HS;
        if (hs_fix != nullptr) {
            const double ff = hs_fix[d], fb = hs_fix[S + d];
            if (ff == ff) fh = ff;
            if (fb == fb) bh = fb;
        }
        of.hs = fh; of.flow = (T0n - t_front_b) * fh;
        ob.hs = bh; ob.flow = (Tnn - t_back_b) * bh;
    }
}

// A small surface's constants and state, fetched once (the general layout: node j of lane l at node_base + j * 64 + l).
template <int CAV>
__device__ __forceinline__ void small_load(const GeneralTile &tile, int lane, const NodeArrays &na, int64_t gen_base,
                                           const SideDyn &df, const SideDyn &db, int nn, Nodes4 &T,
                                           double (&Us)[kSmallNodes], double (&sol)[kSmallNodes], int (&cav)[kSmallNodes],
                                           CavityDev (&cv)[kSmallNodes - 1]) {
    const double *Tg = na.T + tile.node_base + lane;
    const double *Ug = na.U + tile.node_base + lane;
    const int64_t gofs = tile.node_base - gen_base + lane;
    // (every load goes out at once, from the slot of a node the surface has — j clamped —, and is masked afterwards:
    // loads under `j < nn` branches wait for each other, one memory round trip per node)
    double tl[kSmallNodes], ul[kSmallNodes], af[kSmallNodes], ab[kSmallNodes];
    int cl[kSmallNodes];
#pragma unroll
    for (int j = 0; j < kSmallNodes; j++) {
        const int64_t jc = (int64_t)min(j, nn - 1) * kWave;
        tl[j] = Tg[jc];
        ul[j] = Ug[jc];
        af[j] = na.alpha_f[gofs + jc];
        ab[j] = na.alpha_b[gofs + jc];
        cl[j] = CAV ? na.cav[gofs + jc] : -1;
    }
#pragma unroll
    for (int j = 0; j < kSmallNodes; j++) {
        const bool v = j < nn;
        T.set(j, v ? tl[j] : 0.0);
        Us[j] = v ? ul[j] : 0.0;
        cav[j] = (CAV && v) ? cl[j] : -1;
        // surface.rs:930-931
        double sj = v ? af[j] * df.solar : 0.0;
        sj += v ? ab[j] * db.solar : 0.0;
        sol[j] = sj;
    }
    // the cavity records, once (the no-mass loop evaluates Cavity::u_value every pass: surface.rs:814)
#pragma unroll
    for (int j = 0; j < kSmallNodes - 1; j++) {
        cv[j] = CavityDev{0., 0., 0., 0., 0., 0, 0};
        if constexpr (CAV) {
            if (cav[j] >= 0) cv[j] = na.cavs[cav[j]];
        }
    }
}

// One tile of small surfaces (general layout: one lane per surface), one sub-timestep. Returns the lane's passes of
// the no-mass loop.
template <int CAV>
__device__ __forceinline__ unsigned int small_tile_march(int64_t node_base, int surf_base, int G, int lane,
                                                         const NodeArrays &na, int64_t gen_base, const SideArrays &sd,
                                                         const StepWeather &w, const double *__restrict__ zone_T,
                                                         int *__restrict__ flags) {
    if (lane >= G) return 0;
    GeneralTile tile;
    tile.node_base = node_base;
    tile.surf_base = surf_base;
    tile.G = G;
    const int d = surf_base + lane;
    const int S = sd.S;
    const SideConst cf = sd.sc[d];
    const SideConst cb = sd.sc[S + d];
    const SideDyn df = sd.dyn[d];
    const SideDyn db = sd.dyn[S + d];
    const int nn = cf.kind_n >> 16;
    Nodes4 T{0.0, 0.0, 0.0, 0.0};
    double Us[kSmallNodes], sol[kSmallNodes];
    int cav[kSmallNodes];
    CavityDev cv[kSmallNodes - 1];
    small_load<CAV>(tile, lane, na, gen_base, df, db, nn, T, Us, sol, cav, cv);
    int bad = 0;
    unsigned int iters = 0;
    SideOut of, ob;
    small_step<CAV>(cf, cb, df, db, sd.hs_fix, d, S, w, boundary_temperature(cf, w, zone_T),
                    boundary_temperature(cb, w, zone_T), nn, T, Us, sol, cav, cv, bad, iters, of, ob);
    double *Tg = na.T + node_base + lane;
#pragma unroll
    for (int j = 0; j < kSmallNodes; j++) if (j < nn) Tg[(int64_t)j * kWave] = T.get(j);
    sd.out[d] = of;
    sd.out[S + d] = ob;
    {
        const double Tl = T.last(nn);
        put_zone_contrib(sd, cf, of.hs, T.a);
        put_zone_contrib(sd, cb, ob.hs, Tl);
    }
    if (bad) report_failure(flags, bad, (unsigned int)d);
    return iters;
}

// (two wavefronts per SIMD asked for: the cavity variant sits at 260 registers otherwise and runs 100 000 windows in
// two rounds of 1 024 wavefronts instead of one of 1 563)
template <int CAV>
__global__ void __launch_bounds__(256, 2)
k_surfaces_small(const GeneralTile *__restrict__ tiles, int n_tiles, NodeArrays na, int64_t gen_base,
                 SideArrays sd, const CavityDev *__restrict__ cavs,
                 const StepWeather *__restrict__ weather, const int *__restrict__ step_ptr, int step_fixed,
                 const double *__restrict__ zone_T, int *__restrict__ flags,
                 unsigned long long *__restrict__ nomass_iters) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= n_tiles) return;
    const GeneralTile tile = tiles[wave];
    const int step = (step_fixed >= 0) ? step_fixed : *step_ptr;
    const StepWeather w = weather[step];
    const unsigned int iters = small_tile_march<CAV>(tile.node_base, tile.surf_base, tile.G, lane, na, gen_base, sd, w, zone_T, flags);
    if (lane < tile.G) nomass_iters[(int64_t)wave * kWave + lane] += iters;  // one slot per lane of the tile
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------
// Cluster-resident march, pieces shared by the fast-path wavefronts and the small-surface wavefronts of a
// workgroup (layout.hpp, FusedBlock). LDS after the palettes (and V): see k_surfaces_fast.
struct FusedLds {
    double2 *hT;            // (hs * area, face temperature) of the zone-facing sides, in the order of the zones' lists
    double *zT, *za0, *zb0, *zvol;
    double *zsa, *zsb;      // teams: a member's own sums of a zone, between the pass that publishes them and the pass that gathers
    int *zoff;
    unsigned short *slots;  // place in hT of side [2][lanes] (front sides, then back sides, by lane of the workgroup)
};

// Barriers a wavefront of a fused workgroup passes: one at the end of fused_block_init, two per sub-timestep in
// fused_zone_phase. A wavefront without a tile that stays for the work queue keeps step with exactly these counts
// (k_surfaces_fast); a wavefront that leaves instead (no queue) drops out of the barrier's count on gfx9.
constexpr int kFusedBarriersAtInit = 1;
constexpr int kFusedBarriersPerSubstep = 2;

// The block's zone data -> LDS (every wavefront of the block takes part; ends with kFusedBarriersAtInit barrier).
__device__ __forceinline__ void fused_block_init(const FusedBlock &blk, const FusedArgs &fa, const FusedLds &l, int n_threads) {
    const int e_first = fa.zone_eoff[blk.first_zone];
    if ((int)threadIdx.x <= blk.n_zones) l.zoff[threadIdx.x] = fa.zone_eoff[blk.first_zone + threadIdx.x] - e_first;
    if ((int)threadIdx.x < blk.n_zones) {
        const int z = fa.zones[blk.first_zone + threadIdx.x];
        l.zT[threadIdx.x] = fa.zone_T[z];
        l.za0[threadIdx.x] = fa.a0[z];
        l.zb0[threadIdx.x] = fa.b0[z];
        l.zvol[threadIdx.x] = fa.vol[z];
    }
    const int n_e = fa.zone_eoff[blk.first_zone + blk.n_zones] - e_first;
    // where every zone-facing side puts its (hs A, T) pair: at its entry's place in its zone's list, so that the sums
    // below read the pairs in a row (the look-up is the writers' — all wavefronts, side by side — not the summing one's)
    for (int e = threadIdx.x; e < n_e; e += n_threads) l.slots[fa.slots[e_first + e]] = (unsigned short)e;
    __syncthreads();
}

// The same, from values fetched ahead (fast_tile_march, FUSED): a workgroup's first microseconds are a chain of memory
// round trips — block -> tile -> records / T / palettes -> zone list -> zone data — and every workgroup pays it once per march
// call (8.7 % of its life at 20 sub-timesteps per call, 28 % at 5: profiles/r03_fused_phases.txt). The zone list depends on
// the block alone and travels beside the tile descriptor, the zone data beside the tile's own loads.
struct FusedPre {
    int e_first, n_e, zoff, z, slot0;
    double zT, za0, zb0, zvol;
};
__device__ __forceinline__ void fused_pre_a(const FusedBlock &blk, const FusedArgs &fa, FusedPre &p) {
    p = FusedPre{0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0};
    const int nz = blk.n_zones;
    if (nz > 0) {  // (uniform: a block of surfaces that face no zone has no list)
        p.e_first = fa.zone_eoff[blk.first_zone];
        p.n_e = fa.zone_eoff[blk.first_zone + nz];
        p.zoff = fa.zone_eoff[blk.first_zone + min((int)threadIdx.x, nz)];
        p.z = fa.zones[blk.first_zone + min((int)threadIdx.x, nz - 1)];
    }
}
__device__ __forceinline__ void fused_pre_b(const FusedBlock &blk, const FusedArgs &fa, FusedPre &p) {
    if (blk.n_zones > 0) {
        p.n_e -= p.e_first;
        p.zoff -= p.e_first;
        p.zT = fa.zone_T[p.z];
        p.za0 = fa.a0[p.z];
        p.zb0 = fa.b0[p.z];
        p.zvol = fa.vol[p.z];
        if (p.n_e > 0) p.slot0 = fa.slots[p.e_first + min((int)threadIdx.x, p.n_e - 1)];
    }
}
__device__ __forceinline__ void fused_block_init(const FusedBlock &blk, const FusedArgs &fa, const FusedLds &l, int n_threads,
                                                 const FusedPre &p) {
    if ((int)threadIdx.x <= blk.n_zones) l.zoff[threadIdx.x] = p.zoff;
    if ((int)threadIdx.x < blk.n_zones) {
        l.zT[threadIdx.x] = p.zT;
        l.za0[threadIdx.x] = p.za0;
        l.zb0[threadIdx.x] = p.zb0;
        l.zvol[threadIdx.x] = p.zvol;
    }
    if ((int)threadIdx.x < p.n_e) l.slots[p.slot0] = (unsigned short)threadIdx.x;
    for (int e = threadIdx.x + n_threads; e < p.n_e; e += n_threads) l.slots[fa.slots[p.e_first + e]] = (unsigned short)e;
    __syncthreads();
}

// calculate_zones_abc + estimate_zones_future_temperatures for the block's zones (model.rs:489-597,650-674): the sides'
// (hs A, T) pairs are in LDS. A workgroup with no more zones than wavefronts gives every zone a wavefront (k_zones'
// summation tree). One with more zones (buildings: rooms joined by partitions, a dozen walls each) gives every zone
// a ROW of 16 lanes — four zones per wavefront at a time, their sums, divisions and exponentials side by side in the
// lanes instead of one zone after the other: the serial tail of the phase is what the other wavefronts wait for.
// TEAM: the workgroup is a member of a team (layout.hpp): a zone's sums are this member's PARTIAL sums; they are
// published, the other members' awaited (bounded), and all of them added in member order — every member that faces the
// zone gets the same bits.
struct TeamCtx {
    int team, member, round, it;
    int *flags;
};
__device__ __forceinline__ unsigned long long team_granule(unsigned int data, unsigned int tag) {
    return ((unsigned long long)tag << 32) | data;
}
template <int TEAM>
__device__ __forceinline__ void fused_zone_phase(const FusedBlock &blk, const FusedArgs &fa, const FusedLds &l, int wib,
                                                 int n_waves, int lane, int &bad_all, const TeamCtx &tc_) {
    __syncthreads();
    // TEAM: a zone faced from several members is balanced from all their sums. Every member PUBLISHES its sums of ALL its
    // zones before it waits for anybody's (two passes below): a member that waited zone by zone could wait for a sum its
    // partner publishes only after a zone it is itself still to reach — two members that share two zones and meet them in
    // different orders would wait for each other for ever (found by tools/fuzz.py: members of more than 4 x wavefronts zones).
    auto team_area = [&](int j, unsigned int &mask, unsigned int &tag) -> unsigned long long * {
        const unsigned int info = fa.team_zinfo[blk.first_zone + j];
        mask = info >> 16;
        tag = fa.tag_base | ((unsigned int)tc_.round << 12) | (unsigned int)(tc_.it + 1);
        return fa.xbuf + ((((size_t)tc_.team * 2 + (tc_.it & 1)) * kTeamZones + (info & 0xffffu)) * kTeamMax) * 4;
    };
    auto publish = [&](int j, double a, double b) {  // this member's partial sums: four granules, one write-through store each
        unsigned int mask, tag;
        unsigned long long *g = team_area(j, mask, tag) + tc_.member * 4;
        const unsigned long long ab = (unsigned long long)__double_as_longlong(a), bb = (unsigned long long)__double_as_longlong(b);
        __hip_atomic_store(g + 0, team_granule((unsigned int)ab, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 1, team_granule((unsigned int)(ab >> 32), tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 2, team_granule((unsigned int)bb, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 3, team_granule((unsigned int)(bb >> 32), tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto gather = [&](int j, double &a, double &b) {  // the sums of every member that faces the zone, added in member order
        unsigned int mask, tag;
        const unsigned long long *area = team_area(j, mask, tag);
        double sa = 0.0, sb = 0.0;
        for (int m = 0; m < kTeamMax; m++) {
            if (!((mask >> m) & 1u)) continue;
            double am = a, bm = b;
            if (m != tc_.member) {
                const unsigned long long *g = area + m * 4;
                unsigned long long g0 = 0, g1 = 0, g2 = 0, g3 = 0;
                bool got = false;
                for (int spin = 0; spin < (1 << 21); spin++) {  // (~1 us per poll under load: seconds before giving up)
                    g0 = __hip_atomic_load(g + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g1 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g2 = __hip_atomic_load(g + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g3 = __hip_atomic_load(g + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    got = (unsigned int)(g0 >> 32) == tag && (unsigned int)(g1 >> 32) == tag &&
                          (unsigned int)(g2 >> 32) == tag && (unsigned int)(g3 >> 32) == tag;
                    if (got) break;
                    __builtin_amdgcn_s_sleep(4);
                }
                if (!got) atomicOr(tc_.flags, FLAG_EXCHANGE);  // (the member never came: reported, not waited for)
                am = __longlong_as_double((long long)((g0 & 0xffffffffull) | (g1 << 32)));
                bm = __longlong_as_double((long long)((g2 & 0xffffffffull) | (g3 << 32)));
            }
            sa += am;
            sb += bm;
        }
        a = sa;
        b = sb;
    };
    auto finish = [&](int j, double a, double b, double tc, double cz) {
        a += l.za0[j];
        b += l.zb0[j];
        double ft = tc;
        if (fabs(b) > 1e-9) ft = a / b + (tc - a / b) * exp(-b * fa.dt / cz);  // model.rs:662-666
        if (ft != ft) bad_all |= FLAG_NAN_ZONE;                                // model.rs:417-420
        l.zT[j] = ft;
    };
    if (blk.n_zones <= n_waves) {
        if (wib < blk.n_zones) {
            const int j = wib;
            const int e0 = l.zoff[j], e1 = l.zoff[j + 1];
            // (the capacitance does not wait for the sums: its division overlaps their LDS round trips)
            const double tc = l.zT[j];
            const double cz = zone_mcp(l.zvol[j], tc);  // model.rs:549-552
            double a = 0.0, b = 0.0;
            for (int e = e0 + lane; e < e1; e += kWave) {
                const double2 ht = l.hT[e];
                a += ht.x * ht.y;
                b += ht.x;
            }
            a = wave_sum_f64(a);
            b = wave_sum_f64(b);
            if (lane == 0) {
                if constexpr (TEAM) {  // (one zone per wavefront: nothing of this member's is published behind this wait)
                    publish(j, a, b);
                    gather(j, a, b);
                }
                finish(j, a, b, tc, cz);
            }
        }
    } else {
        const int row = lane >> 4, rl = lane & 15;
        const bool one_pass = blk.n_zones <= 4 * n_waves;  // (workgroup-uniform) the loop below runs once
        (void)one_pass;
#pragma clang loop unroll(disable)
        for (int j0 = 0; j0 < blk.n_zones; j0 += 4 * n_waves) {  // (workgroup-uniform trip count)
            const int j = j0 + 4 * wib + row;
            const bool on = j < blk.n_zones;
            const int jj = on ? j : 0;
            const int e0 = l.zoff[jj], e1 = on ? l.zoff[jj + 1] : e0;
            const double tc = l.zT[jj];
            const double cz = zone_mcp(l.zvol[jj], tc);
            double a = 0.0, b = 0.0;
            for (int e = e0 + rl; __any(e < e1); e += 16) {  // (wave-uniform trip count: DPP below must not sit in divergent code)
                if (e < e1) {
                    const double2 ht = l.hT[e];
                    a += ht.x * ht.y;
                    b += ht.x;
                }
            }
            a = row_sum_f64(a);
            b = row_sum_f64(b);
            if constexpr (TEAM) {
                if (on && rl == 15) {
                    publish(j, a, b);
                    if (one_pass) {  // (a row meets one zone only: everything of this member is published before any wait)
                        gather(j, a, b);
                        finish(j, a, b, tc, cz);
                    } else {  // pass 1: keep the own sums for pass 2 (same lane: no barrier needed)
                        l.zsa[j] = a;
                        l.zsb[j] = b;
                    }
                }
            } else {
                if (on && rl == 15) finish(j, a, b, tc, cz);
            }
        }
        if constexpr (TEAM) {
            if (!one_pass) {
#pragma clang loop unroll(disable)
                for (int j0 = 0; j0 < blk.n_zones; j0 += 4 * n_waves) {  // pass 2: everything of this member is published
                    const int j = j0 + 4 * wib + row;
                    if (j < blk.n_zones && rl == 15) {
                        const double tc = l.zT[j];
                        const double cz = zone_mcp(l.zvol[j], tc);
                        double a = l.zsa[j], b = l.zsb[j];
                        gather(j, a, b);
                        finish(j, a, b, tc, cz);
                    }
                }
            }
        }
    }
    __syncthreads();
}

// A wavefront of small surfaces inside a fused workgroup: one lane per surface, everything in registers over
// the march; same arithmetic as k_surfaces_small (small_step), zone temperatures from LDS.
#pragma clang fp contract(off)
__device__ void fused_small_wave(const FusedBlock &blk, const FusedArgs &fa, const FusedLds &l, int lanes_per_side,
                                 int wib, int n_waves, int lane, const NodeArrays &na, const SideArrays &sd,
                                 const StepWeather *__restrict__ weather, int *__restrict__ flags) {
    const int tile_index = blk.first_small + (wib - blk.n_tiles);
    const GeneralTile tile = fa.gen_tiles[tile_index];
    const bool active = lane < tile.G;
    const int d = tile.surf_base + (active ? lane : 0);
    const int S = sd.S;
    const SideConst cf = sd.sc[d];
    const SideConst cb = sd.sc[S + d];
    const SideDyn df = sd.dyn[d];
    const SideDyn db = sd.dyn[S + d];
    const int nn = cf.kind_n >> 16;
    const int lz_f = fa.side_lzone[d], lz_b = fa.side_lzone[S + d];
    const double area = fa.side_area[d];
    Nodes4 T{0.0, 0.0, 0.0, 0.0};
    double Us[kSmallNodes], sol[kSmallNodes];
    int cav[kSmallNodes];
    CavityDev cv[kSmallNodes - 1];
    small_load<1>(tile, active ? lane : 0, na, fa.gen_base, df, db, nn, T, Us, sol, cav, cv);
    fused_block_init(blk, fa, l, n_waves * kWave);
    int bad_all = 0;
    unsigned int iters = 0;
    SideOut of{0.0, 0.0}, ob{0.0, 0.0};
    auto btemp = [&](const SideConst &c, int lz, const StepWeather &w) -> double {  // model.rs:79-96
        const int kind = c.kind_n & 3;
        if (kind == KIND_SPACE) return l.zT[lz];
        if (kind == KIND_AMBIENT) return c.ambient;
        return w.t_out;
    };
#pragma clang loop unroll(disable)
    for (int it = 0; it < fa.n_sub; it++) {
        const StepWeather w = weather[it];
        int bad = 0;
        small_step<1>(cf, cb, df, db, sd.hs_fix, d, S, w, btemp(cf, lz_f, w), btemp(cb, lz_b, w), nn, T, Us, sol, cav,
                      cv, bad, iters, of, ob);
        if (active) {
            bad_all |= bad;
            const double Tl = T.last(nn);
            if ((cf.kind_n & 3) == KIND_SPACE) l.hT[l.slots[wib * kWave + lane]] = make_double2(of.hs * area, T.a);
            if ((cb.kind_n & 3) == KIND_SPACE) l.hT[l.slots[lanes_per_side + wib * kWave + lane]] = make_double2(ob.hs * area, Tl);
        }
        fused_zone_phase<0>(blk, fa, l, wib, n_waves, lane, bad_all, TeamCtx{0, 0, 0, 0, nullptr});
    }
    if ((int)threadIdx.x < blk.n_zones) fa.zone_T[fa.zones[blk.first_zone + threadIdx.x]] = l.zT[threadIdx.x];
    if (active) {
        double *Tg = na.T + tile.node_base + lane;
#pragma unroll
        for (int j = 0; j < kSmallNodes; j++) if (j < nn) Tg[(int64_t)j * kWave] = T.get(j);
        sd.out[d] = of;
        sd.out[S + d] = ob;
        if (fa.small_iters) fa.small_iters[(int64_t)tile_index * kWave + lane] += iters;
    }
    if (bad_all) report_failure(flags, bad_all, (unsigned int)d);  // (lane 0 of a zone-owning wave may carry a zone flag while inactive)
}
#pragma clang fp contract(fast)

// Diagnostic build only (heat_amd/build.py build_stamps, -DHEAT_STAMPS -> lib/libheat_amd_stamps.so; tools/fused_phases.py):
// wavefront 0 of every cluster-resident workgroup stamps the shader clock at its phases and the 100 MHz real-time
// clock at both ends into a buffer nothing else reads. In the product build no stamp exists.
#ifdef HEAT_STAMPS
constexpr int kStampsPerBlock = 8;
__device__ unsigned long long g_stamps[65536 * kStampsPerBlock];
#define HEAT_STAMP(k, real)                                                                                   \
    do {                                                                                                      \
        if constexpr (FUSED) {                                                                                \
            if (threadIdx.x == 0 && (counter_index >> 2) < 65536)                                             \
                g_stamps[(counter_index >> 2) * kStampsPerBlock + (k)] =                                      \
                    (real) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();                 \
        }                                                                                                     \
    } while (0)
#else
#define HEAT_STAMP(k, real) do { } while (0)
#endif
// Diagnostic build only (-DHEAT_STREAM_STAMPS -> lib/libheat_amd_sstamps.so; tools/stream_phases.py): lane 0 of every
// STREAMED fast tile stamps the shader clock at the tile's phases — 0 start, 1 loads arrived, 6 boundary terms / no-mass
// loop done, 7 RK4 done, 2 new convection coefficients and contributions done, 3 stores acknowledged — slot = tile index
// (+ 32768 for tiles of 16 nodes per lane: the wide part has a list of its own).
#ifdef HEAT_STREAM_STAMPS
#ifndef HEAT_STAMPS
constexpr int kStampsPerBlock = 8;
__device__ unsigned long long g_stamps[65536 * kStampsPerBlock];
#endif
#define HEAT_SSTAMP(k, drain)                                                                                  \
    do {                                                                                                       \
        if constexpr (!FUSED && PAL) {                                                                         \
            if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                             \
            if (lane == 0 && counter_index < 32768)                                                            \
                g_stamps[(counter_index + (M == 16 ? 32768 : 0)) * kStampsPerBlock + (k)] = __builtin_amdgcn_s_memtime(); \
        }                                                                                                      \
    } while (0)
#else
#define HEAT_SSTAMP(k, drain) do { } while (0)
#endif

// One tile of a fast class: n_it sub-timesteps (one, unless FUSED) of its surfaces. The body of k_surfaces_fast and
// of the fast-path cases of k_surfaces_stream.
//   write_out      streamed: hs and the heat flows (model.rs:154-169) are observable after the LAST sub-timestep of a
//                  march call only — the launches before it leave those 32 bytes per surface unwritten
//   counter_index  slot of the tile in `nomass_iters` (NM)
//   nm_on          NM variants: whether this tile holds walls with no-mass facings at all (wave-uniform; the unified
//                  streamed kernel runs all-massive and faced tiles through one variant)
template <int M, int NM, int PAL, int CAV, int FUSED, int SMALL = 0>
__device__ __forceinline__ void fast_tile_march(const FastTile tile, int counter_index, bool nm_on, int lane, int wib,
                                                double *s_pal, double *s_V, double *s_pos, const FusedLds &fl, const FusedBlock &blk,
                                                int blk_waves, int n_it, int step0, const NodeArrays &na,
                                                const SideArrays &sd, const StepWeather *__restrict__ weather,
                                                const double *__restrict__ zone_T, int *__restrict__ flags,
                                                unsigned long long *__restrict__ nomass_iters, const FusedArgs &fa,
                                                bool write_out = true, const TeamCtx team = TeamCtx{0, 0, 0, 0, nullptr}) {
    constexpr int kLanes = (FUSED ? FUSED : 4) * kWave;
    constexpr bool kVinLds = FUSED && M == 16;
    HEAT_STAMP(4, true);
    HEAT_STAMP(0, false);
    HEAT_SSTAMP(0, false);
    double2 *const s_hT = fl.hT;
    double *const s_zT = fl.zT;
    (void)s_hT; (void)s_zT; (void)s_V; (void)s_pos; (void)nm_on; (void)blk; (void)blk_waves; (void)counter_index;
    FusedPre pre;
    (void)pre;
    if constexpr (FUSED) fused_pre_a(blk, fa, pre);  // (beside the tile descriptor: both depend on the block alone)
    const bool mixed = PAL && (tile.k & kTileMixedBit) != 0;  // (wave-uniform) surfaces of different lane counts
    // (wave-uniform) the tile holds no-mass chunks other than one-node facings: chunks inside the wall, of two nodes
    // (streamed variants only: the cluster-resident march leaves clusters with such walls to the streamed kernels — the
    // chunk loop's registers would cost every fused NM variant, used or not)
    // NM = 2 compiles that chunk loop in; NM = 1 knows one-node facings only (its face terms live in ten registers
    // fewer: what puts the 8-node streamed body under the 168 registers of three wavefronts per SIMD)
    constexpr bool kChunks = NM == 2 && PAL;
    const bool chunky = kChunks && (tile.k & kTileChunkyBit) != 0;
    const int k = tile.k & 0xff;                              // lanes per surface; mixed: lanes of the tile
    const bool full = (tile.k & 0x100) != 0;
    const int G = tile.G;
    const int Lk = mixed ? k : (kWave / k) * k;
    int g = (int)(((float)lane + 0.5f) * (1.0f / (float)k));
    int seg = lane - g * k;
    bool is_last = (seg == k - 1);
    const bool in_layout = lane < Lk;
    if (mixed) {
        // the lane's surface and segment from the tile's lane table (behind its class bytes, layout.hpp)
        const unsigned int e = reinterpret_cast<const unsigned short *>(na.cls + tile.node_base + (int64_t)M * Lk)[in_layout ? lane : 0];
        g = (int)(e & 63u);
        seg = (int)((e >> 6) & 63u);
        is_last = (e & (unsigned int)kLaneLastBit) != 0;
    }
    const bool active = in_layout && (g < G);
    if (!active) g = 0;
    const int d = tile.surf_base + g;
    const int ll = in_layout ? lane : 0;

    // ---- side record of this lane: the first lane of a surface owns the front side, every other
    // lane loads the back side (only the last lane's is used). k == 1: both sides, see below. ----
    const int S = sd.S;
    const bool is_first = (seg == 0);
    // A surface of ONE lane owns both its sides on it (front as every first lane does, then the back). The
    // cluster-resident march carries that path for 8 nodes per lane without gas cavities only (elsewhere the second
    // side's registers cost a wavefront per SIMD or spill: those workgroups hold surfaces of two lanes or more — the
    // host sees to it).
    constexpr bool kSingleLane = !FUSED || (M == 8 && !CAV);
    const bool single = kSingleLane && is_first && is_last;
    const bool my_back = !is_first;
    const int sidx = (my_back ? S : 0) + d;
    // Every load of the tile is issued here, from addresses that depend on the tile descriptor alone: a record that
    // names another record (the second side of a one-lane wall, the front side behind a back / Ambient one) would put
    // a memory round trip of its own in front of the tile — the streamed kernels are bound by exactly that chain.
    const SideConst c_load = sd.sc[sidx];
    const SideDyn dy_load = sd.dyn[sidx];
    SideConst cb2 = c_load;
    SideDyn db2 = dy_load;
    if constexpr (kSingleLane) {
        // (uniform tiles: a one-lane tile is one for every lane — a scalar branch, nothing waits for it)
        if ((!mixed && k == 1) || single) {  // this lane is also the last one of its surface
            cb2 = sd.sc[S + d];
            db2 = sd.dyn[S + d];
        }
    }
    int4 cavref = make_int4(-1, -1, -1, -1);
    if constexpr (CAV) cavref = reinterpret_cast<const int4 *>(na.cavref)[d];
    const int kind_n_mine = c_load.kind_n;
    // streamed: a Space-facing side's place in its zone's contribution list waits in LDS for the end of the tile (in
    // two registers across the RK stages it costs the 16-node variant its second wavefront per SIMD; fetched again
    // from L2 at the end it is a dependent load in front of every tile's last store)
    if constexpr (!FUSED) s_pos[threadIdx.x] = c_load.ambient;
    int my_lz = 0, b_lz = 0;
    double my_area = 0.0;
    if constexpr (FUSED) {
        my_lz = fa.side_lzone[sidx];
        my_area = fa.side_area[sidx];
        if constexpr (kSingleLane) b_lz = single ? fa.side_lzone[S + d] : 0;
    }
    (void)b_lz;
    // FUSED: block-local zone of the FRONT side behind this back side (a back / Ambient side reads t_front, below)
    int peer_lz = 0;
    if constexpr (FUSED) peer_lz = fa.side_lzone[d];
    (void)peer_lz;

    // ---- node data: T, V = dt/C, U (coalesced 16-byte loads) ----
    double T[M], V[M], U[M];
    unsigned int cw[M / 4];  // PAL: the lane's class bytes, four per word
    (void)cw;
    if constexpr (PAL) {
        const int pstride = na.pal_stride;  // doubles per palette (layout.hpp)
        const int ubase = na.pal_ubase;
        double *sp = s_pal + wib * (kWave * pstride);
        auto load_cls_T = [&]() {
            const unsigned char *pc = na.cls + tile.node_base + (int64_t)ll * M;
            if constexpr (M == 4) {
                cw[0] = *reinterpret_cast<const unsigned int *>(pc);
            } else {
#pragma unroll
                for (int q = 0; q < M / 8; q++) {
                    const uint2 w0 = reinterpret_cast<const uint2 *>(pc)[q];
                    cw[2 * q] = w0.x;
                    cw[2 * q + 1] = w0.y;
                }
            }
            const double2 *pT = reinterpret_cast<const double2 *>(na.T + tile.node_base);
#pragma unroll
            for (int jp = 0; jp < M / 2; jp++) {
                const double2 t = pT[jp * Lk + ll];
                T[2 * jp] = t.x; T[2 * jp + 1] = t.y;
            }
        };
        // FUSED: the whole of a workgroup's start in flight at once — class bytes and T first, the palette pieces pinned
        // where they are issued (the compiler sinks each load under the guard of its LDS write: six load-wait-write
        // round trips), the zone data behind them. A resident workgroup starts into an idle memory system, its start is
        // pure latency (13 000 of its 150 000 ticks at 20 sub-timesteps per call before this, 10 000 after). NOT for the
        // streamed kernels: their tiles wait in a saturated memory pipeline, and the same order cost them registers for
        // nothing (profiles/experiments/r03_all_tile_loads_in_flight.patch).
        if constexpr (FUSED) {
            load_cls_T();
            fused_pre_b(blk, fa, pre);
        }
        {   // the palettes of this tile's surfaces: G * pstride contiguous doubles -> LDS
            const double2 *gp = reinterpret_cast<const double2 *>(na.pal + (int64_t)tile.surf_base * pstride);
            double2 *sp2 = reinterpret_cast<double2 *>(sp);
            const int n2 = G * (pstride / 2);
            // (six loads in flight, then six LDS writes: a load-wait-write loop costs a memory round trip per pass)
            for (int i0 = 0; i0 < n2; i0 += 6 * kWave) {
                double2 t6[6];
#pragma unroll
                for (int q = 0; q < 6; q++) t6[q] = gp[min(i0 + q * kWave + lane, n2 - 1)];
                if constexpr (FUSED) {
                    asm volatile("" : "+v"(t6[0].x), "+v"(t6[1].x), "+v"(t6[2].x), "+v"(t6[3].x), "+v"(t6[4].x), "+v"(t6[5].x),
                                      "+v"(pre.zT), "+v"(pre.za0), "+v"(pre.zb0), "+v"(pre.zvol), "+v"(pre.slot0));
                }
#pragma unroll
                for (int q = 0; q < 6; q++)
                    if (i0 + q * kWave + lane < n2) sp2[i0 + q * kWave + lane] = t6[q];
            }
        }
        if constexpr (!FUSED) load_cls_T();
        HEAT_STAMP(6, false);
        __builtin_amdgcn_wave_barrier();  // LDS writes above are ordered before the reads below (same wave)
        {
            const double *mp = sp + g * pstride;
#pragma unroll
            for (int j = 0; j < M; j++) {
                const unsigned int cbj = (cw[j >> 2] >> (8 * (j & 3))) & 0xff;
                const unsigned int vi = cbj & (kPalV - 1);
                V[j] = mp[vi >= kPalVMark1 ? 0u : vi];  // (14, 15: no-mass chunk marks, see below; entry 0 is 0.0)
                U[j] = mp[ubase + ((cbj >> kPalUShift) & (kPalU - 1))];
            }
        }
        if constexpr (kVinLds) {
#pragma unroll
            for (int j = 0; j < M; j++) s_V[j * kLanes + threadIdx.x] = V[j];
        }
    } else {
        const double2 *pT = reinterpret_cast<const double2 *>(na.T + tile.node_base);
        const double2 *pV = reinterpret_cast<const double2 *>(na.V + tile.node_base);
        const double2 *pU = reinterpret_cast<const double2 *>(na.U + tile.node_base);
#pragma unroll
        for (int jp = 0; jp < M / 2; jp++) {
            const double2 t = pT[jp * Lk + ll];
            const double2 v = pV[jp * Lk + ll];
            const double2 u = pU[jp * Lk + ll];
            T[2 * jp] = t.x; T[2 * jp + 1] = t.y;
            V[2 * jp] = v.x; V[2 * jp + 1] = v.y;
            U[2 * jp] = u.x; U[2 * jp + 1] = u.y;
        }
    }
    HEAT_STAMP(7, false);
    if constexpr (FUSED) fused_block_init(blk, fa, fl, blk_waves * kWave, pre);  // the block's zone data -> LDS

    const int first_lane = lane - seg;
    const int nn = kind_n_mine >> 16;
    // local index of the last node inside the last lane (the last lane's segment number is its surface's k - 1)
    const int jl = full ? (M - 1) : (nn - 1 - (mixed ? seg : k - 1) * M);

    auto pick_last = [&](const double (&x)[M]) {
        double r = x[M - 1];
        if (!full) {
#pragma unroll
            for (int j = 0; j < M - 1; j++) r = (j == jl) ? +x[j] : +r;
        }
        return r;
    };

    // does any back side of this tile face an ambient temperature? (wave-uniform)
    // (the other side's kind travels in every record: bits 4-5 of kind_n)
    const bool wave_quirk = __any((is_last && (kind_n_mine & 3) == KIND_AMBIENT && my_back) ||
                                  (single && ((kind_n_mine >> 4) & 3) == KIND_AMBIENT));
    // streamed: the zone temperatures of this lane's side(s), gathered as soon as the records are here — for every
    // lane (SideConst::zone is always a valid index), so that no branch sits between the records and the gather
    double zt_mine = 0.0, zt_b2 = 0.0;
    (void)zt_mine; (void)zt_b2;
    if constexpr (!FUSED) {
        zt_mine = zone_T[c_load.zone];
        if constexpr (kSingleLane) zt_b2 = zone_T[cb2.zone];
    }

    // V = dt/C of local node j: registers, or the workgroup's LDS array (FUSED with 16 nodes per lane)
    auto Vat = [&](int j) -> double {
        if constexpr (kVinLds) return s_V[j * kLanes + threadIdx.x];
        else return V[j];
    };
    auto V_last = [&]() {
        double r = Vat(M - 1);
        if (!full) {
#pragma unroll
            for (int j = 0; j < M - 1; j++) r = (j == jl) ? Vat(j) : r;
        }
        return r;
    };

    int bad_all = 0;
    unsigned int nm_passes = 0;
    double o_hs = 0.0, o_flow = 0.0, o2_hs = 0.0, o2_flow = 0.0;  // outputs of the last sub-timestep
    StepWeather w_next = weather[step0];

    HEAT_STAMP(1, false);
    HEAT_SSTAMP(1, true);
    // The side record: FUSED fetches it again for every sub-timestep (an L1/L2 hit) instead of holding its 20 registers
    // across the RK stages — at the END of the sub-timestep before, so that it travels while the zone balance is summed.
    SideConst c_cur = c_load;
    SideDyn dy_cur = dy_load;
#pragma clang loop unroll(disable)
    for (int it = 0; it < n_it; it++) {  // sub-timesteps (one, unless FUSED)
    const StepWeather w = w_next;
    if constexpr (FUSED) w_next = weather[min(it + 1, n_it - 1)];  // fetched a whole sub-timestep ahead of its use
    const SideConst c = c_cur;
    const SideDyn dy = dy_cur;
    // get_boundary_temperature, model.rs:79-96 (FUSED: zone temperatures live in LDS)
    // zt: the gathered temperature of cc.zone (streamed)
    auto btemp = [&](const SideConst &cc, int lz, double zt) -> double {
        const int kind = cc.kind_n & 3;
        if (kind == KIND_SPACE) {
            if constexpr (FUSED) return s_zT[lz];
            else return zt;
        }
        if (kind == KIND_AMBIENT) return cc.ambient;
        return w.t_out;
    };

    int bad = 0;
    // The part of calc_border_conditions (surface.rs:596-717) that does not change inside a sub-timestep:
    // boundary temperature, radiant temperature, forced convection term, and whether the side reads the
    // FRONT surface temperature (back/Ambient takes t_front and the front temperature, surface.rs:672-686).
    // Kept in plain scalars (a struct here ends up in scratch memory).
    auto prepare = [&](const SideConst &cc, const SideDyn &dd, bool back, int rec, int lz, double zt, double &air_t,
                       double &rad_t, double &forced, bool &use_front_T) {
        const int kind = cc.kind_n & 3;
        air_t = btemp(cc, lz, zt);
        rad_t = air_t;
        use_front_T = false;
        forced = 0.0;
        if (kind == KIND_OUTDOOR) {
            const bool windward = (cc.kind_n & 4) ? true : ((cc.nx * w.sin_wd + cc.ny * w.cos_wd) > 0.0);
            forced = (windward ? 1.0 : 0.5) * (cc.forced * w.sqrt_ws);  // convection.rs:161-163
            rad_t = dd.rad_t;                                            // surface.rs:647,692
        } else if (back && kind == KIND_AMBIENT) {
            // t_front (surface.rs:672-686): the front side's boundary source travels in this record (layout.hpp) —
            // its kind in bits 4-5, its zone in `zone` (whose temperature is zt), its ambient temperature in `forced`
            const int fkind = (cc.kind_n >> 4) & 3;
            if (fkind == KIND_SPACE) {
                if constexpr (FUSED) rad_t = s_zT[peer_lz];
                else rad_t = zt;
            } else if (fkind == KIND_AMBIENT) {
                rad_t = cc.forced;
            } else {
                rad_t = w.t_out;
            }
            use_front_T = true;
        }
        (void)rec;
    };
    // (the debug override of a side, surface.rs:708-714, is read where it is applied — a uniform branch on a
    // pointer that is null in production — instead of living in two registers across the RK stages)
    auto conv = [&](double air_t, double forced, double nat_pos, double nat_neg, int rec, double surf_t) {
        double hs = forced + tarp_natural_coef(air_t, surf_t, nat_pos, nat_neg, bad);  // convection.rs:165-167
        if (hs != hs) bad |= FLAG_NAN_HS;                                // surface.rs:704-707
        if (sd.hs_fix != nullptr) {
            const double fix = sd.hs_fix[rec];
            if (fix == fix) hs = fix;
        }
        return hs;
    };

    // Surface temperatures as the reference reads them from `state` (pre-step). A side reads its own face node,
    // which its owner lane holds; only a back side facing an ambient temperature reads the FRONT node
    // (surface.rs:672-686) and needs it from the surface's first lane.
    const double T0 = wave_quirk ? shfl_f64(T[0], first_lane) : T[0];
    const double Tn = pick_last(T);

    // Conductance towards the previous lane's last node.
    double UL = from_prev_lane(U[M - 1]);
    if (is_first) UL = 0.0;
    // Last node of the previous lane (wave-wide exchange: must not sit inside a divergent branch).
    double T_prev_last = 0.0;
    if constexpr (NM) T_prev_last = from_prev_lane(T[M - 1]);  // (wave-wide: not under the nm_on gate of a lane)

    // Face terms of get_k_q (discretization.rs:658-697) + solar gains (surface.rs:916-931,766-769).
    double hF = 0.0, qF = 0.0, hB = 0.0, qB = 0.0;
    // add_face returns the face conductance and source through h_out / q_out (the caller files them
    // under front or back with selects: storing through a runtime side index lands in scratch).
    // (NM: the terms of a face are kept for the no-mass chunk that may sit at it, below)
    double fF_hs = 0.0, fF_rhs = 0.0, fF_rad = 0.0, fF_air = 0.0, fF_sol = 0.0;  // front face of this lane's surface
    double fB_hs = 0.0, fB_rhs = 0.0, fB_rad = 0.0, fB_air = 0.0, fB_sol = 0.0;  // back face
    (void)fF_hs; (void)fF_rhs; (void)fF_rad; (void)fF_air; (void)fF_sol;
    (void)fB_hs; (void)fB_rhs; (void)fB_rad; (void)fB_air; (void)fB_sol;
    auto add_face = [&](const SideConst &cc, const SideDyn &dd, bool back, double air_t, double rad_t, double forced,
                        int rec, bool use_front_T, double &h_out, double &q_out) {
        h_out = 0.0;
        q_out = 0.0;
        const double surf_t = (back && !use_front_T) ? Tn : T0;
        const double hs = conv(air_t, forced, cc.cos_eff, cc.alpha, rec, surf_t);  // (fast classes: the two coefficients)
        const double rhs = rad_hs(cc.emis, rad_t, surf_t);  // surface.rs:941-948
        const double sol = dd.solar;                        // absorbed: alpha * irradiance, formed at upload
        if constexpr (NM) {
            if constexpr (kChunks) {
                fF_hs = back ? fF_hs : hs; fF_rhs = back ? fF_rhs : rhs; fF_rad = back ? fF_rad : rad_t;
                fF_air = back ? fF_air : air_t; fF_sol = back ? fF_sol : sol;
                fB_hs = back ? hs : fB_hs; fB_rhs = back ? rhs : fB_rhs; fB_rad = back ? rad_t : fB_rad;
                fB_air = back ? air_t : fB_air; fB_sol = back ? sol : fB_sol;
            }
            // a no-mass face node is (part of) a chunk; V == 0: it takes no part in the RK4
            const double vface = back ? V_last() : Vat(0);
            if (nm_on && active && vface == 0.0 && nn >= 2) {
                if (chunky) return;  // solved with the tile's other chunks below
                // The common case, a tile whose only chunks are one-node facings — solved right here
                // (discretization.rs:658-697 for nnodes == 1):
                //   K = (0 - h_face) - u_inner,  q = (q_face + u_inner * T_inner) + solar,  x = -q / K,
                //   T <- (T + x) / 2 until the error stops shrinking or err < tol (surface.rs:836-895).
                double u_in, t_in;
                if (!back) {
                    u_in = U[0];
                    t_in = T[1 % M];
                } else {
                    // inner neighbour of the last node: previous node of this lane, or of the previous lane
                    double up = UL, tp = T_prev_last;
#pragma unroll
                    for (int j = 1; j < M; j++) {
                        up = (j == jl) ? +U[j - 1] : +up;
                        tp = (j == jl) ? +T[j - 1] : +tp;
                    }
                    u_in = up;
                    t_in = tp;
                }
                double Tc = back ? pick_last(T) : T[0];
                const double dg = (0.0 - hs) - u_in;
                const double nb = u_in * t_in;
                double old_err = 99999.;
                int count = 0;
                for (;;) {
                    const double qf = air_t * hs + rhs * (rad_t - Tc);
                    const double q = ((back ? (nb + qf) : (qf + nb)) + sol) * -1.;
                    const double x = q / dg;
                    const double err = fabs(x - Tc);
                    nm_passes++;
                    if (err > old_err) break;                            // surface.rs:842-848
                    if (err != err) { bad |= FLAG_NAN_NOMASS; break; }   // surface.rs:850
                    Tc = (Tc + x) * 0.5;
                    const double tol = (count < 100) ? 0.01 : 0.5;      // surface.rs:885
                    if (err < tol) break;
                    old_err = err;
                    count++;
                }
                if (!back) {
                    T[0] = Tc;
                } else {
#pragma unroll
                    for (int j = 0; j < M; j++) T[j] = (j == jl) ? Tc : T[j];
                }
                return;
            }
        }
        const double tface = back ? pick_last(T) : T[0];
        h_out = hs;
        q_out = (air_t * hs + rhs * (rad_t - tface)) + sol;
    };

    double my_air, my_rad, my_forced;
    bool my_useF;
    prepare(c, dy, my_back, sidx, my_lz, zt_mine, my_air, my_rad, my_forced, my_useF);
    if (is_first || is_last) {
        double h_, q_;
        add_face(c, dy, my_back, my_air, my_rad, my_forced, sidx, my_useF, h_, q_);
        hF = my_back ? 0.0 : h_;
        qF = my_back ? 0.0 : q_;
        hB = my_back ? h_ : 0.0;
        qB = my_back ? q_ : 0.0;
    }
    double b_air = 0.0, b_forced = 0.0, b_cos = 0.0, b_neg = 0.0;
    bool b_useF = false;
    if constexpr (kSingleLane) {
        if (single) {  // this lane is also the last one of its surface
            double b_rad;
            prepare(cb2, db2, true, S + d, b_lz, zt_b2, b_air, b_rad, b_forced, b_useF);
            b_cos = cb2.cos_eff;
            b_neg = cb2.alpha;
            add_face(cb2, db2, true, b_air, b_rad, b_forced, S + d, b_useF, hB, qB);
        }
    }

    if constexpr (kChunks) {
        // ---- no-mass chunks (march_nomass, surface.rs:790-898), before the massive nodes march (surface.rs:950-965) ----
        // A chunk is one or two consecutive no-mass nodes between massive nodes and / or a face: a thin facing, two
        // light layers at a face (render on insulation), an insulation layer and an air gap inside a cavity wall. Its
        // nodes sit in one lane; the class byte of its first node says so (V index 14 or 15: 1 or 2 nodes). Tiles whose only
        // chunks are one-node facings (the common case, and all that per-node-array classes know) have solved them
        // above; a tile marked kTileChunkyBit solves ALL its chunks here. Every pass rebuilds the chunk's K and q as get_k_q does
        // (discretization.rs:596-700: interior segments, then the front term, then the back term), solves K x = -q
        // (mut_n_diag_gaussian) and halves the distance, T <- (T + x) / 2, with the reference's exit rules.
        const double T_next_first = from_next_lane(T[0]);  // (wave-wide exchange, outside the divergent code)
        unsigned int starts = 0;                            // two bits per local node
        if (nm_on && chunky && active) {
#pragma unroll
            for (int j = 0; j < M; j++) {
                const unsigned int vi = (cw[j >> 2] >> (8 * (j & 3))) & (kPalV - 1);
                starts |= (vi >= kPalVMark1 ? vi - (kPalVMark1 - 1) : 0u) << (2 * j);
            }
        }
        while (__any(starts != 0)) {
            if (starts != 0) {
                const int j0 = (__ffs((int)starts) - 1) >> 1;
                const int cn = (int)((starts >> (2 * j0)) & 3u);  // nodes of the chunk: 1 or 2
                starts &= ~(3u << (2 * j0));
                const int j1 = j0 + cn - 1;                       // its last node
                // operands: the chunk's temperatures, its neighbours' (a lane further where the chunk touches the
                // lane's end), the conductances around and inside it
                double ta = T[0], tb = T[0], tp = T_prev_last, tn = T_next_first, up = UL, ui = 0.0, un = 0.0;
#pragma unroll
                for (int j = 0; j < M; j++) {
                    ta = (j == j0) ? +T[j] : +ta;
                    tb = (j == j1) ? +T[j] : +tb;
                    if (j + 1 < M) tp = (j + 1 == j0) ? +T[j] : +tp;
                    if (j > 0) tn = (j == j1 + 1) ? +T[j] : +tn;
                    if (j + 1 < M) up = (j + 1 == j0) ? +U[j] : +up;
                    ui = (j == j0) ? +U[j] : +ui;
                    un = (j == j1) ? +U[j] : +un;
                }
                const bool faceL = is_first && j0 == 0;           // the chunk starts at the front face
                const bool faceR = is_last && j1 == jl;           // ... ends at the back face
                const double hL = faceL ? fF_hs : up, hR = faceR ? fB_hs : un;
                const double solA = faceL ? fF_sol : 0.0;
                const double solB = faceR ? fB_sol : 0.0;         // (cn == 1 at the back face: added to node a)
                double old_err = 99999.;
                int count = 0;
                for (;;) {
                    const double t_last = (cn == 2) ? tb : ta;
                    const double qL = faceL ? (fF_air * fF_hs + fF_rhs * (fF_rad - ta)) : (up * tp);
                    const double qR = faceR ? (fB_air * fB_hs + fB_rhs * (fB_rad - t_last)) : (un * tn);
                    double x0, x1 = 0.0, err;
                    if (cn == 1) {
                        const double dg = (0.0 + -hL) + -hR;
                        const double q = (((0.0 + qL) + qR) + (solA + solB)) * -1.;
                        x0 = q / dg;
                        err = fabs(x0 - ta);
                    } else {
                        const double u = ui;
                        double dg0 = 0.0 + -u, dg1 = 0.0 - u;
                        const double up0 = 0.0 + u, lo1 = 0.0 + u;
                        dg0 += -hL;
                        dg1 += -hR;
                        const double q0 = ((0.0 + qL) + solA) * -1.;
                        double q1 = ((0.0 + qR) + solB) * -1.;
                        const double f = lo1 / dg0;                // mut_n_diag_gaussian(q, 3)
                        dg1 -= f * up0;
                        q1 -= f * q0;
                        x1 = q1 / dg1;
                        x0 = (q0 - up0 * x1) / dg0;
                        err = fabs(x0 - ta) + fabs(x1 - tb);
                    }
                    nm_passes++;
                    if (err > old_err) break;                            // surface.rs:842-848
                    if (err != err) { bad |= FLAG_NAN_NOMASS; break; }   // surface.rs:850
                    ta = (ta + x0) * 0.5;
                    if (cn == 2) tb = (tb + x1) * 0.5;
                    const double tol = (count < 100) ? 0.01 : 0.5;      // surface.rs:885
                    if (err / (double)cn < tol) break;
                    old_err = err;
                    count++;
                }
#pragma unroll
                for (int j = 0; j < M; j++) {
                    T[j] = (j == j0) ? ta : T[j];
                    if (j > 0) T[j] = (cn == 2 && j == j1) ? tb : T[j];
                }
            }
        }
    }

    if constexpr (CAV) {
        const double T_next_first = from_next_lane(T[0]);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int node = r ? cavref.z : cavref.x;
            const int cidx = r ? cavref.w : cavref.y;
            const int jc = node - seg * M;
            if (active && cidx >= 0 && jc >= 0 && jc < M) {
                double ta = T[0], tb = T_next_first;
#pragma unroll
                for (int j = 0; j < M; j++) {
                    ta = (j == jc) ? +T[j] : +ta;
                    if (j + 1 < M) tb = (j == jc) ? +T[j + 1] : +tb;
                }
                const double u = cavity_u_value(na.cavs[cidx], ta, tb, bad);
#pragma unroll
                for (int j = 0; j < M; j++) U[j] = (j == jc) ? u : U[j];
            }
        }
        UL = from_prev_lane(U[M - 1]);
        if (is_first) UL = 0.0;
    }

    // ---- RK4 (surface.rs:228-308) ----
    {
        // effective conductances of the two faces (apply_A): once per sub-timestep, not per stage and node
        const double hBe = is_last ? hB : 0.0;
        if (full) {
            U[M - 1] = is_last ? hB : U[M - 1];
        } else {
#pragma unroll
            for (int j = 0; j < M; j++) U[j] = (j == jl) ? U[j] + hBe : U[j];  // (the last node's own U is 0: UValue::Back)
        }
        const double ULe = is_first ? hF : UL;
        const double qFe = is_first ? qF : 0.0;
        const double qBe = is_last ? qB : 0.0;
#ifdef HEAT_STREAM_STAMPS
        asm volatile("" : "+v"(T[0]), "+v"(T[M - 1]));
        HEAT_SSTAMP(6, false);
#endif
        if (full) {
            rk4_horner<M>(T, Vat, U, ULe, is_first, is_last, qFe, [&](int j) -> double { return (j == M - 1) ? qBe : 0.0; });
        } else {
            rk4_horner<M>(T, Vat, U, ULe, is_first, is_last, qFe, [&](int j) -> double { return (j == jl) ? qBe : 0.0; });
        }
#ifdef HEAT_STREAM_STAMPS
        asm volatile("" : "+v"(T[0]), "+v"(T[M - 1]));
        HEAT_SSTAMP(7, false);
#endif
        if constexpr (FUSED) {  // U lives on: back to UValue::Back at the last node
            if (full) {
                U[M - 1] = is_last ? 0.0 : U[M - 1];
            } else {
#pragma unroll
                for (int j = 0; j < M; j++) U[j] = (is_last && j == jl) ? 0.0 : U[j];
            }
        }
    }

    // ---- write back node temperatures (model.rs:145-147); FUSED: after the last sub-timestep only ----
    if constexpr (!FUSED) {
        if (active) {
            double2 *pT = reinterpret_cast<double2 *>(na.T + tile.node_base);
#pragma unroll
            for (int jp = 0; jp < M / 2; jp++) pT[jp * Lk + lane] = make_double2(T[2 * jp], T[2 * jp + 1]);
        }
    }

    // ---- convection coefficients with the NEW temperatures + heat flows (model.rs:150-169) ----
    const double T0n = wave_quirk ? shfl_f64(T[0], first_lane) : T[0];
    const double Tln = pick_last(T);
    const double Tnn = Tln;
    {
        const double surf_t = (my_back && !my_useF) ? Tnn : T0n;
        const double hs = conv(my_air, my_forced, c.cos_eff, c.alpha, sidx, surf_t);
        const double face_t = my_back ? Tln : T[0];
        o_hs = hs;
        o_flow = (face_t - my_air) * hs;
        if constexpr (!FUSED) {
            // this side's share of its zone's heat balance (its place in the zone's list: from LDS, see above)
            if (active && (is_first || is_last) && (kind_n_mine & 3) == KIND_SPACE) {
                ZoneContrib z;
                z.ha = hs * c.forced;  // (the area: layout.hpp, SideConst::forced of a Space-facing side)
                z.t_face = face_t;
                sd.zc[__double_as_longlong(s_pos[threadIdx.x])] = z;
            }
        }
        if constexpr (FUSED) {
            if (active && (is_first || is_last) && (c.kind_n & 3) == KIND_SPACE)
                s_hT[fl.slots[(my_back ? kLanes : 0) + wib * kWave + lane]] = make_double2(hs * my_area, face_t);
        }
    }
    if (single) {
        const double hs = conv(b_air, b_forced, b_cos, b_neg, S + d, b_useF ? T0n : Tnn);
        o2_hs = hs;
        o2_flow = (Tln - b_air) * hs;
        if constexpr (!FUSED) {
            if (active) put_zone_contrib(sd, cb2, hs, Tln);
        } else {
            if (active && (cb2.kind_n & 3) == KIND_SPACE)
                s_hT[fl.slots[kLanes + wib * kWave + lane]] = make_double2(hs * my_area, Tln);
        }
    } else if (!(is_first || is_last)) {
        bad = 0;
    }
    bad_all |= bad;

    if constexpr (FUSED) {
        int sidx_next = sidx;
        asm volatile("" : "+v"(sidx_next));  // (a fresh load, not the value of the pass before kept in registers)
        c_cur = sd.sc[sidx_next];
        dy_cur = sd.dyn[sidx_next];
        if constexpr (SMALL == 2) {
            TeamCtx tcx = team;
            tcx.it = it;
            fused_zone_phase<1>(blk, fa, fl, wib, blk_waves, lane, bad_all, tcx);
        } else {
            fused_zone_phase<0>(blk, fa, fl, wib, blk_waves, lane, bad_all, team);
        }
    }
    }  // sub-timesteps
    HEAT_STAMP(2, false);
    HEAT_SSTAMP(2, false);

    if constexpr (FUSED) {
        if (active) {
            double2 *pT = reinterpret_cast<double2 *>(na.T + tile.node_base);
#pragma unroll
            for (int jp = 0; jp < M / 2; jp++) pT[jp * Lk + lane] = make_double2(T[2 * jp], T[2 * jp + 1]);
        }
        if (threadIdx.x < blk.n_zones) fa.zone_T[fa.zones[blk.first_zone + threadIdx.x]] = s_zT[threadIdx.x];
    }
    if (write_out && active && (is_first || is_last)) {
        SideOut o;
        o.hs = o_hs;
        o.flow = o_flow;
        sd.out[sidx] = o;
    }
    if (write_out && single && active) {
        SideOut o;
        o.hs = o2_hs;
        o.flow = o2_flow;
        sd.out[S + d] = o;
    }
    if (active && bad_all) report_failure(flags, bad_all, (unsigned int)d);
#ifdef HEAT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stores acknowledged: what a retiring wavefront waits for)
#endif
    HEAT_STAMP(3, false);
    HEAT_SSTAMP(3, true);
    HEAT_STAMP(5, true);
    if constexpr (NM) {
        // passes of the no-mass loop, summed per tile (one owner per slot: no atomics on a shared word)
        unsigned int tot = nm_passes;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) tot += __shfl_down(tot, o, kWave);
        if (lane == 0 && tot) nomass_iters[counter_index] += tot;
    }
}

// NM = 1: the surface may carry a no-mass FACING node (node 0 and/or node n-1, every other node
// massive): each is a one-node no-mass chunk, solved by the reference's damped fixed-point loop
// (march_nomass, surface.rs:790-898) in the face lane before the massive nodes march.
// PAL = 1: V and U come from the surface's palette (staged in LDS) through one class byte per node
// instead of two doubles per node (layout.hpp).
// CAV = 1: up to two gas cavities between massive nodes; their conductance (Cavity::u_value, cavity.rs:59-69)
// is evaluated once per sub-timestep from the temperatures the massive chunk starts from, as get_k_q does
// (discretization.rs:634-639), and frozen over the four RK stages (surface.rs:268-293).
// FUSED = 1: cluster-resident march (layout.hpp, FusedBlock). One workgroup holds every surface facing its
// zones; it marches fa.n_sub sub-timesteps of ThermalModel::march (model.rs:369-424) in one launch: the node
// temperatures never leave the registers, the zone balance (calculate_zones_abc + the analytic update,
// model.rs:489-597,650-674) is summed from LDS in the same order and with the same arithmetic as k_zones, and
// only the final temperatures, coefficients and flows are written back.
// (FUSED is the most wavefronts a workgroup may hold: 4 or 8; it needs PAL = 1; with cavities, M <= 8.)
// SMALL = 1: the workgroup may also hold wavefronts of small all-no-mass surfaces (fused_small_wave).
template <int M, int NM, int PAL, int CAV, int FUSED, int SMALL = 0>
__global__ void __launch_bounds__(FUSED ? 64 * FUSED : 256, (FUSED || (M == 16 && !CAV)) ? 2 : 1)
k_surfaces_fast(const FastTile *__restrict__ tiles, int n_tiles, NodeArrays na, SideArrays sd,
                const StepWeather *__restrict__ weather, const int *__restrict__ step_ptr, int step_fixed,
                const double *__restrict__ zone_T, int *__restrict__ flags,
                unsigned long long *__restrict__ nomass_iters, FusedArgs fa) {
    static_assert(!FUSED || (PAL && !(CAV && M == 16)),
                  "the cluster-resident march exists for palette classes; with gas cavities for 4 or 8 nodes per lane");
    constexpr int kMaxW = FUSED ? FUSED : 4;
    constexpr int kLanes = kMaxW * kWave;
    // LDS: the palettes of the block's tiles; FUSED adds the per-side (hs, face temperature) pairs the zone
    // balance is summed from, [2][kLanes] double2, and the zone temperatures (dynamic: > 64 KB for 8 waves).
    extern __shared__ double s_dyn[];
    __shared__ double s_pos_static[FUSED ? 2 : kLanes];
    double *const s_pal = s_dyn;  // (dynamic: kLanes * na.pal_stride doubles)
    // FUSED, after the palettes: with 16 nodes per lane V = dt/C of every node, [M][kLanes] (read where it is used:
    // held in registers over the march it would push that variant out of the register file); (hs * area, face temperature)
    // per side [2][kLanes] double2; zone temperatures, a0, b0, volume [kFusedMaxZones] each; first slot of every
    // zone [kFusedMaxZones + 1]; the slot lists.
    constexpr bool kVinLds = FUSED && M == 16;
    double *const s_V = s_dyn + kLanes * na.pal_stride;
    FusedLds fl;
    fl.hT = reinterpret_cast<double2 *>(s_V + (kVinLds ? M : 0) * kLanes);
    fl.zT = reinterpret_cast<double *>(fl.hT) + 4 * kLanes;
    fl.za0 = fl.zT + kFusedMaxZones;
    fl.zb0 = fl.za0 + kFusedMaxZones;
    fl.zvol = fl.zb0 + kFusedMaxZones;
    fl.zsa = fl.zvol + kFusedMaxZones;
    fl.zsb = fl.zsa + kFusedMaxZones;
    fl.zoff = reinterpret_cast<int *>(fl.zsb + kFusedMaxZones);
    fl.slots = reinterpret_cast<unsigned short *>(fl.zoff + kFusedMaxZones + 2);
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x >> 6;
    int wave0 = blockIdx.x * (blockDim.x >> 6) + wib;
    int n_waves = gridDim.x * (blockDim.x >> 6);
    FusedBlock blk{0, 0, 0, 0, 0, 0};
    int blk_waves = 0;  // wavefronts of the workgroup that have work
    int bi = blockIdx.x;  // FUSED: the FusedBlock this workgroup marches
    __shared__ int s_next_block;
    const int n_it = FUSED ? fa.n_sub : 1;
    const int step0 = FUSED ? 0 : ((step_fixed >= 0) ? step_fixed : *step_ptr);
    // (step_ptr[1]: index of the last sub-timestep of the running march call; a caller that steps by hand gets every output)
    const bool write_out = FUSED || step_fixed >= 0 || step0 == step_ptr[1];
    // Streaming (FUSED = 0), persistent waves: each walks the tile list with a grid stride, so the write-back of
    // one tile (a wave cannot retire before its stores are acknowledged) overlaps the loads of the next.
    // FUSED: one pass per FusedBlock; with a work queue (sharded batches, fa.queue) the workgroup takes further ones.
    // TEAM (SMALL == 2): the workgroup is member blockIdx.x % team_size of team blockIdx.x / team_size; the team walks the
    // clusters (FusedSuper) team, team + n_teams, ... — every member in the same order, so that the members of a cluster
    // are always at work on it together (layout.hpp).
    constexpr bool kTeam = FUSED && SMALL == 2;
    TeamCtx team{0, 0, 0, 0, flags};
    int sb = 0, n_teams = 1;
    if constexpr (kTeam) {
        team.team = (int)blockIdx.x / fa.team_size;
        team.member = (int)blockIdx.x % fa.team_size;
        n_teams = (int)gridDim.x / fa.team_size;
        sb = team.team;
    }
    for (int wave = wave0; FUSED ? (kTeam ? (sb < fa.n_super) : (bi < fa.n_blocks)) : (wave < n_tiles); wave += n_waves) {
    if constexpr (FUSED) {
        if constexpr (kTeam) {
            const FusedSuper su = fa.supers[fa.reverse ? fa.n_super - 1 - sb : sb];  // (every member maps alike)
            if (team.member >= su.n_members) {  // (the whole workgroup: this cluster has fewer members than a team)
                sb += n_teams;
                team.round++;
                continue;
            }
            bi = su.first_block + team.member;
        }
        // (reverse: the list from its end — consecutive march calls take the workgroups' clusters in opposite orders, so
        // that the temperatures the last call wrote last are read first, out of the memory-side cache; team members
        // are named by their cluster and stay as they are)
        blk = fa.blocks[(!kTeam && fa.reverse) ? fa.n_blocks - 1 - bi : bi];
        blk_waves = blk.n_tiles + (SMALL == 1 ? blk.n_small : 0);
        const bool looping = kTeam || fa.queue != nullptr;  // the workgroup goes on to another block after this one
        if (wib >= blk_waves) {
            // no tile for this wavefront. One FusedBlock per workgroup: done (a finished wave does not take part in
            // s_barrier). A workgroup that goes on to another block keeps the wavefront, in step with the barriers.
            if (!looping) return;
            // kFusedBarriersAtInit + kFusedBarriersPerSubstep * n_sub: the same count as a working wavefront passes
            // (fused_block_init ends with one barrier, fused_zone_phase brackets the zone sums with two)
            for (int q = 0; q < kFusedBarriersAtInit; q++) __syncthreads();
            for (int it = 0; it < fa.n_sub; it++)
                for (int q = 0; q < kFusedBarriersPerSubstep; q++) __syncthreads();
            goto next_block;
        }
        if constexpr (SMALL == 1) {
            if (wib >= blk.n_tiles) {
                fused_small_wave(blk, fa, fl, kLanes, wib, blk_waves, lane, na, sd, weather, flags);
                if (!looping) return;
                goto next_block;
            }
        }
        wave = blk.first_tile + wib;
    }
    // (streamed: consecutive sub-timesteps walk the list in opposite directions — what the last sweep touched last is
    // what this one reads first, out of the memory-side cache instead of HBM)
    {
    const int tix = (!FUSED && fa.reverse) ? n_tiles - 1 - wave : wave;
    fast_tile_march<M, NM, PAL, CAV, FUSED, SMALL>(tiles[tix], tix, true, lane, wib, s_pal, s_V, s_pos_static, fl, blk, blk_waves, n_it, step0, na,
                                                   sd, weather, zone_T, flags, nomass_iters, fa, write_out, team);
    }
next_block:
    if constexpr (FUSED) {
        if constexpr (kTeam) {
            __syncthreads();  // every wavefront is done with this block's LDS
            sb += n_teams;
            team.round++;
        } else {
            if (fa.queue == nullptr) break;
            __syncthreads();  // every wavefront is done with this block's LDS
            if (threadIdx.x == 0) s_next_block = (int)(gridDim.x + atomicAdd(fa.queue, 1u));
            __syncthreads();
            bi = s_next_block;
        }
    }
    }  // tile / block loop
}

// ---------------------------------------------------------------------------
// The streamed sub-timestep in ONE launch (iterate_surfaces, model.rs:102-180, over every surface that is not
// marched cluster-resident): persistent wavefronts walk a single tile list that holds the palette-form fast classes
// of every blocking factor and the cavity-free small surfaces; FastTile::k carries the tile's kind —
//   bits 9-10  0 / 1 / 2: fast-path tile of 4 / 8 / 16 nodes per lane; 3: small all-no-mass surfaces, one per lane
//   bit 11     the tile holds walls with no-mass facings (8 and 4 nodes per lane only: the 16-node variant with
//              facings does not fit two wavefronts per SIMD)
// — a wave-uniform switch instead of one launch per class: no class has a tail of its own, the latency-bound
// small tiles are spread through the list and disappear behind the streaming ones. Classes this kernel does not
// hold (per-node constants, gas cavities, the catch-all) keep their own launches.
constexpr int kTileKindShift = 9;
constexpr int kTileNmBit = 1 << 11;
// The kernel comes in three variants, each under the launch bounds its bodies can keep, because a wavefront of this
// latency-bound walk is parked at s_waitcnt for most of its life (SQ_WAIT_ANY 59-63 % at two wavefronts per SIMD):
//   kStreamWide   tiles of 16 nodes per lane (no facings): 2 wavefronts per SIMD (T, v, z, U, V alone are 160 registers)
//   kStreamLight  tiles of 8 / 4 nodes per lane with one-node facings at most, and the small surfaces: 3 wavefronts
//                 per SIMD (<= 168 registers)
//   kStreamChunks tiles of 8 / 4 nodes per lane that hold other no-mass chunks (inside the wall, of two nodes): the
//                 chunk loop's registers, 2 wavefronts per SIMD
//   kStreamCav    everything with a gas cavity that takes 8 / 4 nodes per lane or one lane per surface: double glazing
//                 (small surfaces whose no-mass loop re-evaluates Cavity::u_value every pass, surface.rs:814) and walls
//                 with cavities between massive nodes (Trombe walls) — both bound by the latency of the Nusselt
//                 correlations' transcendental chains; in ONE launch (the windows first: theirs are the longest tiles)
//                 instead of two on forked streams, whose fork and join cost more than either kernel's tail
// The host sorts a batch's tiles into one list per variant (batch.hip, rebuild_unified) and launches them back to
// back on the batch's stream.
enum { kStreamWide = 0, kStreamLight = 1, kStreamChunks = 2, kStreamCav = 3 };
template <int VARIANT>
__global__ void __launch_bounds__(256, VARIANT == kStreamLight ? 3 : 2)
k_surfaces_stream(const FastTile *__restrict__ tiles, int n_tiles, NodeArrays na, int64_t gen_base, SideArrays sd,
                  const StepWeather *__restrict__ weather, const int *__restrict__ step_ptr, int step_fixed,
                  const double *__restrict__ zone_T, int *__restrict__ flags,
                  unsigned long long *__restrict__ nomass_iters, int reverse) {
    extern __shared__ double s_pal[];  // 4 * kWave * na.pal_stride doubles
    __shared__ double s_pos[4 * kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x >> 6;
    const int wave0 = blockIdx.x * (blockDim.x >> 6) + wib;
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    const int step0 = (step_fixed >= 0) ? step_fixed : *step_ptr;
    const bool write_out = step_fixed >= 0 || step0 == step_ptr[1];
    const FusedLds fl{};
    const FusedBlock blk{0, 0, 0, 0, 0, 0};
    const FusedArgs fa{};
    if (wave0 >= n_tiles) return;
    constexpr int kNm = VARIANT == kStreamChunks ? 2 : 1;
    // (the tile descriptor is wave-uniform: scalar loads; the next one is fetched a whole tile ahead of its use)
    // reverse: the list from its end — consecutive sub-timesteps sweep the batch in opposite directions, so that what the
    // last sweep touched last (still in the 256 MB memory-side cache) is what this one reads first
    const int last = n_tiles - 1;
    FastTile tile_next = tiles[__builtin_amdgcn_readfirstlane(reverse ? last - wave0 : wave0)];
    for (int wv = wave0; wv < n_tiles; wv += n_waves) {
        const int w = __builtin_amdgcn_readfirstlane(reverse ? last - wv : wv);
        FastTile tile = tile_next;
        const int wn = min(wv + n_waves, last);
        tile_next = tiles[__builtin_amdgcn_readfirstlane(reverse ? last - wn : wn)];
        const int kind = (tile.k >> kTileKindShift) & 3;
        const bool nm = (tile.k & kTileNmBit) != 0;
        tile.k = (int16_t)(tile.k & (0x1ff | kTileMixedBit | kTileChunkyBit));
        if constexpr (VARIANT == kStreamWide) {
            fast_tile_march<16, 0, 1, 0, 0>(tile, w, false, lane, wib, s_pal, nullptr, s_pos, fl, blk, 0, 1, step0, na, sd, weather,
                                            zone_T, flags, nomass_iters, fa, write_out);
        } else {
            constexpr int kCav = VARIANT == kStreamCav ? 1 : 0;
            switch (kind) {
            case 1:
                fast_tile_march<8, kNm, 1, kCav, 0>(tile, w, nm, lane, wib, s_pal, nullptr, s_pos, fl, blk, 0, 1, step0, na, sd, weather,
                                                    zone_T, flags, nomass_iters, fa, write_out);
                break;
            case 0:
                fast_tile_march<4, kNm, 1, kCav, 0>(tile, w, nm, lane, wib, s_pal, nullptr, s_pos, fl, blk, 0, 1, step0, na, sd, weather,
                                                    zone_T, flags, nomass_iters, fa, write_out);
                break;
            default: {
                unsigned int tot = small_tile_march<kCav>(tile.node_base, tile.surf_base, tile.G, lane, na, gen_base, sd, weather[step0],
                                                          zone_T, flags);
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) tot += __shfl_down(tot, o, kWave);
                if (lane == 0 && tot) nomass_iters[w] += tot;
                break;
            }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// General path: one lane per surface, reference operation order, no FMA contraction.
#pragma clang fp contract(off)
__global__ void __launch_bounds__(256)
k_surfaces_general(const GeneralTile *__restrict__ tiles, int n_tiles, NodeArrays na, int64_t gen_base,
                   SideArrays sd, const CavityDev *__restrict__ cavs, double *__restrict__ scratch,
                   const StepWeather *__restrict__ weather, const int *__restrict__ step_ptr, int step_fixed,
                   const double *__restrict__ zone_T, int *__restrict__ flags,
                   unsigned long long *__restrict__ nomass_iters) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= n_tiles) return;
    const GeneralTile tile = tiles[wave];
    if (lane >= tile.G) return;
    const int d = tile.surf_base + lane;
    const int S = sd.S;

    const SideConst cf = sd.sc[d];
    const SideConst cb = sd.sc[S + d];
    const SideDyn df = sd.dyn[d];
    const SideDyn db = sd.dyn[S + d];
    const int nn = cf.kind_n >> 16;
    const int bk = cb.kind_n & 3;
    const int step = (step_fixed >= 0) ? step_fixed : *step_ptr;
    const StepWeather w = weather[step];

    double *Tg = na.T + tile.node_base + lane;                       // T(i) = Tg[i*64]
    const double *Vg = na.V + tile.node_base + lane;
    const double *Ug = na.U + tile.node_base + lane;
    const int64_t gofs = tile.node_base - gen_base + lane;
    const double *Fa = na.alpha_f + gofs;
    const double *Ba = na.alpha_b + gofs;
    const double *Mg = na.mass + gofs;
    const int32_t *Cg = na.cav + gofs;
    const int nmax = tile.n_max;
    double *Sx = scratch + tile.scratch_base + lane;                 // SC(a, i) = Sx[(a*nmax + i)*64]
#define SC(a, i) Sx[((int64_t)(a) * nmax + (i)) * kWave]
#define TT(i) Tg[(int64_t)(i) * kWave]
    enum { LO = 0, DG = 1, UP = 2, QQ = 3, AUX = 4, KN = 5, ACC = 6 };

    int bad = 0;
    const double t_front_b = boundary_temperature(cf, w, zone_T);
    const double t_back_b = boundary_temperature(cb, w, zone_T);

    // calc_border_conditions on the pre-step state (surface.rs:596-717): identical for every
    // call made before the write-back at model.rs:145-147.
    const double T0 = TT(0), Tn = TT(nn - 1);
    const bool quirk = (bk == KIND_AMBIENT);  // back/Ambient: t_front and the FRONT temperature (surface.rs:672-686)
    double f_hs, f_rad, b_hs, b_rad;
    const double f_surf = T0;
    const double b_surf = quirk ? T0 : Tn;
    eval_side(cf, w, t_front_b, t_front_b, df.rad_t, f_surf, f_hs, f_rad, bad);
    eval_side(cb, w, t_back_b, quirk ? t_front_b : t_back_b, db.rad_t, b_surf, b_hs, b_rad, bad);
    if (f_hs != f_hs || b_hs != b_hs) bad |= FLAG_NAN_HS;
    if (sd.hs_fix != nullptr) {
        const double ff = sd.hs_fix[d], fb = sd.hs_fix[S + d];
        if (ff == ff) f_hs = ff;
        if (fb == fb) b_hs = fb;
    }
    const double f_air = t_front_b, b_air = t_back_b;
    const double f_radhs = rad_hs(cf.emis, f_rad, f_surf);  // surface.rs:941-948
    const double b_radhs = rad_hs(cb.emis, b_rad, b_surf);

    const double solar_front = df.solar, solar_back = db.solar;  // clamped at upload (surface.rs:916-923)
    auto solar = [&](int i) {  // surface.rs:930-931
        double s = Fa[(int64_t)i * kWave] * solar_front;
        s += Ba[(int64_t)i * kWave] * solar_back;
        return s;
    };
    auto uval = [&](int gidx, double ta, double tb) {  // UValue::u_value, discretization.rs:48-55
        const int cidx = Cg[(int64_t)gidx * kWave];
        if (cidx >= 0) return cavity_u_value(cavs[cidx], ta, tb, bad);
        return Ug[(int64_t)gidx * kWave];
    };
    // Discretization::get_k_q (discretization.rs:596-700) into LO/DG/UP/QQ
    auto get_k_q = [&](int ini, int fin) {
        const int nc = fin - ini;
        for (int li = 0; li < nc; li++) { SC(LO, li) = 0.0; SC(DG, li) = 0.0; SC(UP, li) = 0.0; SC(QQ, li) = 0.0; }
        for (int li = 0; li < nc - 1; li++) {
            const int gi = ini + li;
            const double u = uval(gi, TT(gi), TT(gi + 1));
            SC(DG, li) += -u;
            SC(DG, li + 1) = SC(DG, li + 1) - u;
            SC(UP, li) = SC(UP, li) + u;
            SC(LO, li + 1) = SC(LO, li + 1) + u;
        }
        double hf, fq;
        if (ini == 0) {
            const double ts = TT(0);
            fq = f_air * f_hs + f_radhs * (f_rad - ts);
            hf = f_hs;
        } else {
            const double tb = TT(ini - 1), ta = TT(ini);
            const double u = uval(ini - 1, tb, ta);
            hf = u;
            fq = u * tb;
        }
        SC(QQ, 0) += fq;
        SC(DG, 0) += -hf;
        double hb, bq;
        if (fin == nn) {
            const double ts = TT(fin - 1);
            bq = b_air * b_hs + b_radhs * (b_rad - ts);
            hb = b_hs;
        } else {
            const double tb = TT(fin - 1), ta = TT(fin);
            const double u = uval(fin - 1, tb, ta);
            hb = u;
            bq = u * ta;
        }
        SC(QQ, nc - 1) += bq;
        SC(DG, nc - 1) += -hb;
    };
    auto matvec = [&](int li, int nc, auto xfn) {  // Matrix::prod_tri_diag_into, one row
        double a = 0.0;
        if (li > 0) a += SC(LO, li) * xfn(li - 1);
        a += SC(DG, li) * xfn(li);
        if (li < nc - 1) a += SC(UP, li) * xfn(li + 1);
        return a;
    };

    // ---- no-mass chunks first (surface.rs:950-965, march_nomass :790-898) ----
    unsigned long long iters = 0;
    for (int i = 0; i < nn;) {
        if (Mg[(int64_t)i * kWave] >= kMassThreshold) { i++; continue; }
        const int ini = i;
        while (i < nn && Mg[(int64_t)i * kWave] < kMassThreshold) i++;
        const int fin = i, nc = fin - ini;
        double old_err = 99999.;
        int count = 0;
        for (;;) {
            get_k_q(ini, fin);
            iters++;
            for (int li = 0; li < nc; li++) SC(QQ, li) = (SC(QQ, li) + solar(ini + li)) * -1.;
            // mut_n_diag_gaussian(q, 3): elimination + back-substitution; x overwrites QQ
            for (int li = 1; li < nc; li++) {
                const double f = SC(LO, li) / SC(DG, li - 1);
                SC(DG, li) -= f * SC(UP, li - 1);
                SC(QQ, li) -= f * SC(QQ, li - 1);
            }
            SC(QQ, nc - 1) = SC(QQ, nc - 1) / SC(DG, nc - 1);
            for (int li = nc - 2; li >= 0; li--) SC(QQ, li) = (SC(QQ, li) - SC(UP, li) * SC(QQ, li + 1)) / SC(DG, li);
            double err = 0.0;
            for (int li = 0; li < nc; li++) err += fabs(SC(QQ, li) - TT(ini + li));
            if (err > old_err) break;                         // surface.rs:842-848
            if (err != err) { bad |= FLAG_NAN_NOMASS; break; }  // surface.rs:850
            for (int li = 0; li < nc; li++) TT(ini + li) = (TT(ini + li) + SC(QQ, li)) * 0.5;
            const double tol = (count < 100) ? 0.01 : 0.5;  // surface.rs:885
            if (err / (double)nc < tol) break;
            old_err = err;
            count++;
        }
    }

    // ---- massive chunks (surface.rs:984-1000, march_mass :720-787, rk4 :228-308) ----
    for (int i = 0; i < nn;) {
        if (Mg[(int64_t)i * kWave] < kMassThreshold) { i++; continue; }
        const int ini = i;
        while (i < nn && Mg[(int64_t)i * kWave] >= kMassThreshold) i++;
        const int fin = i, nc = fin - ini;
        get_k_q(ini, fin);
        for (int li = 0; li < nc; li++) {
            const double v = Vg[(int64_t)(ini + li) * kWave];  // dt / C (host-computed, same IEEE division)
            SC(QQ, li) = (SC(QQ, li) + solar(ini + li)) * v;
            if (li > 0) SC(LO, li) *= v;
            SC(DG, li) *= v;
            if (li < nc - 1) SC(UP, li) *= v;
        }
        auto xT = [&](int li) { return TT(ini + li); };
        auto xA = [&](int li) { return SC(AUX, li); };
        for (int li = 0; li < nc; li++) SC(KN, li) = matvec(li, nc, xT) + SC(QQ, li);
        for (int li = 0; li < nc; li++) {
            const double k1 = SC(KN, li), t = TT(ini + li);
            SC(AUX, li) = k1 * 0.5 + t;
            SC(ACC, li) = t + k1 / 6.;
        }
        for (int li = 0; li < nc; li++) SC(KN, li) = matvec(li, nc, xA) + SC(QQ, li);
        for (int li = 0; li < nc; li++) {
            const double k2 = SC(KN, li), t = TT(ini + li);
            SC(AUX, li) = k2 * 0.5 + t;
            SC(ACC, li) += k2 / 3.;
        }
        for (int li = 0; li < nc; li++) SC(KN, li) = matvec(li, nc, xA) + SC(QQ, li);
        for (int li = 0; li < nc; li++) {
            const double k3 = SC(KN, li), t = TT(ini + li);
            SC(AUX, li) = k3 + t;
            SC(ACC, li) += k3 / 3.;
        }
        for (int li = 0; li < nc; li++) SC(KN, li) = matvec(li, nc, xA) + SC(QQ, li);
        for (int li = 0; li < nc; li++) TT(ini + li) = SC(ACC, li) + SC(KN, li) / 6.;
    }

    // ---- outputs (model.rs:150-169): coefficients with the new surface temperatures ----
    {
        const double T0n = TT(0), Tnn = TT(nn - 1);
        double fh, bh, r_;
        eval_side(cf, w, t_front_b, t_front_b, df.rad_t, T0n, fh, r_, bad);
        eval_side(cb, w, t_back_b, t_back_b, db.rad_t, quirk ? T0n : Tnn, bh, r_, bad);
        if (fh != fh || bh != bh) bad |= FLAG_NAN_HS;
        if (sd.hs_fix != nullptr) {
            const double ff = sd.hs_fix[d], fb = sd.hs_fix[S + d];
            if (ff == ff) fh = ff;
            if (fb == fb) bh = fb;
        }
        SideOut of, ob;
        of.hs = fh; of.flow = (T0n - t_front_b) * fh;
        ob.hs = bh; ob.flow = (Tnn - t_back_b) * bh;
        sd.out[d] = of;
        sd.out[S + d] = ob;
        put_zone_contrib(sd, cf, fh, T0n);
        put_zone_contrib(sd, cb, bh, Tnn);
    }
    if (bad) report_failure(flags, bad, (unsigned int)d);
    if (iters) nomass_iters[(int64_t)wave * kWave + lane] += iters;  // one slot per lane of the tile
#undef SC
#undef TT
}
#pragma clang fp contract(fast)


// ---------------------------------------------------------------------------
// Zones: one wavefront per zone (ROWS = 0), or — zones of a few walls each, buildings of small rooms — one ROW of 16
// lanes per zone, four zones per wavefront (ROWS = 1: the sums, divisions and exponentials of four zones side by side).
//   mode 0: every zone, full update (single GPU).
//   mode 1: every zone, write the partial (a, b) into partial[2][n_zones] only.
//   mode 3: as mode 0, but only the zones in zlist[n_list] (the others are owned by fused workgroups).
//   mode 2: sharded: only the zones in zlist[n_list] (those this rank's surfaces touch); a zone no other rank
//           touches (slot_of[z] < 0) is updated here and now, a shared one writes its partial (a, b) into the
//           compact partial[2][n_shared] at its slot for the exchange.
template <int ROWS>
__global__ void __launch_bounds__(256)
k_zones(const int64_t *__restrict__ zone_off, const ZoneEntry *__restrict__ entries,
        const ZoneContrib *__restrict__ zc,
        const double *__restrict__ a0, const double *__restrict__ b0, const double *__restrict__ zone_vol,
        double *__restrict__ zone_T, double *__restrict__ partial, int n_zones, double dt,
        int *__restrict__ step_ptr, int *__restrict__ flags, int mode, const int32_t *__restrict__ zlist,
        int n_list, const int32_t *__restrict__ slot_of, int n_shared) {
    constexpr int kW = ROWS ? 16 : kWave;  // lanes per zone
    const int lane = threadIdx.x & (kW - 1);
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) / kW;
    if (blockIdx.x == 0 && threadIdx.x == 0 && (mode == 0 || mode == 3)) *step_ptr += 1;
    int z = wv;
    if (mode == 2 || mode == 3) {
        if (wv >= n_list) return;  // (whole rows leave: the row sums below stay inside a row)
        z = zlist[wv];
    }
    if (z >= n_zones) return;
    double a = 0.0, b = 0.0;
    (void)entries;
    const int64_t e0 = zone_off[z], e1 = zone_off[z + 1];
    // the zone's own terms are fetched beside the first entries (their latency is then off the serial tail)
    const double za0 = a0[z], zb0 = b0[z], tc = zone_T[z], zv = zone_vol[z];
    // the surface kernels have left every side's (hs, face temperature) at its place in the zone's list: two
    // contiguous streams, two entries per lane and pass
    for (int64_t e = e0 + lane; e < e1; e += 2 * kW) {  // model.rs:562-585
        const bool two = e + kW < e1;
        const int64_t e2 = two ? e + kW : e;
        const ZoneContrib c0 = zc[e], c1 = zc[e2];
        a += c0.ha * c0.t_face;
        b += c0.ha;
        if (two) {
            a += c1.ha * c1.t_face;
            b += c1.ha;
        }
    }
    if constexpr (ROWS) {
        // (the rows' loops above may differ in length: exec is whole again here, rows that left stay masked)
        a = row_sum_f64(a);  // fixed tree: run-to-run deterministic; the row's sum in its lane 15
        b = row_sum_f64(b);
        if (lane != 15) return;
    } else {
        a = wave_sum_f64(a);
        b = wave_sum_f64(b);
        if (lane != 0) return;
    }
    if (mode == 1) {
        partial[z] = a;
        partial[n_zones + z] = b;
        return;
    }
    if (mode == 2) {
        const int slot = slot_of[z];
        if (slot >= 0) {
            partial[slot] = a;
            partial[n_shared + slot] = b;
            return;
        }
    }
    a += za0;
    b += zb0;
    const double c = zone_mcp(zv, tc);  // model.rs:549-552
    double ft = tc;
    if (fabs(b) > 1e-9) ft = a / b + (tc - a / b) * exp(-b * dt / c);  // model.rs:662-666
    if (ft != ft) report_failure(flags, FLAG_NAN_ZONE, (unsigned int)z);  // model.rs:417-420
    zone_T[z] = ft;
}

// Zone update from n_blocks gathered partial blocks ([block][2][n_zones]), summed in block order.
__global__ void __launch_bounds__(256)
k_zone_update(const double *__restrict__ gathered, int n_blocks, const double *__restrict__ a0,
              const double *__restrict__ b0, const double *__restrict__ zone_vol,
              double *__restrict__ zone_T, int n_zones, double dt, int *__restrict__ step_ptr,
              int *__restrict__ flags) {
    const int z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z == 0) *step_ptr += 1;
    if (z >= n_zones) return;
    double a = a0[z], b = b0[z];
    for (int r = 0; r < n_blocks; r++) {
        a += gathered[(int64_t)r * 2 * n_zones + z];
        b += gathered[(int64_t)r * 2 * n_zones + n_zones + z];
    }
    const double tc = zone_T[z];
    const double c = zone_mcp(zone_vol[z], tc);
    double ft = tc;
    if (fabs(b) > 1e-9) ft = a / b + (tc - a / b) * exp(-b * dt / c);
    if (ft != ft) report_failure(flags, FLAG_NAN_ZONE, (unsigned int)z);
    zone_T[z] = ft;
}

// Shared zones of a sharded batch: gathered = [block][2][n_shared]; zone ids in shared_zone[n_shared].
__global__ void __launch_bounds__(256)
k_zone_update_shared(const double *__restrict__ gathered, int n_blocks, const int32_t *__restrict__ shared_zone,
                     int n_shared, const double *__restrict__ a0, const double *__restrict__ b0,
                     const double *__restrict__ zone_vol, double *__restrict__ zone_T, double dt,
                     int *__restrict__ flags) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_shared) return;
    const int z = shared_zone[s];
    double a = a0[z], b = b0[z];
    for (int r = 0; r < n_blocks; r++) {  // rank order: every replica of the zone gets the same bits
        a += gathered[(int64_t)r * 2 * n_shared + s];
        b += gathered[(int64_t)r * 2 * n_shared + n_shared + s];
    }
    const double tc = zone_T[z];
    const double c = zone_mcp(zone_vol[z], tc);
    double ft = tc;
    if (fabs(b) > 1e-9) ft = a / b + (tc - a / b) * exp(-b * dt / c);
    if (ft != ft) report_failure(flags, FLAG_NAN_ZONE, (unsigned int)z);
    zone_T[z] = ft;
}

// ---------------------------------------------------------------------------
// State transfer between the caller's flat SimulationState mirror and the device layout.
template <int M>
__global__ void __launch_bounds__(256)
k_nodes_fast(const FastTile *__restrict__ tiles, int n_tiles, double *__restrict__ Tbuf,
             const int32_t *__restrict__ meta, const int64_t *__restrict__ first_slot,
             double *__restrict__ state, int to_state, const uint8_t *__restrict__ cls) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= n_tiles) return;
    const FastTile tile = tiles[wave];
    const bool mixed = (tile.k & kTileMixedBit) != 0;
    const int k = tile.k & 0xff;
    const int Lk = mixed ? k : (kWave / k) * k;
    int g = (int)(((float)lane + 0.5f) * (1.0f / (float)k));
    int seg = lane - g * k;
    if (lane >= Lk) return;
    if (mixed) {
        const unsigned int e = reinterpret_cast<const unsigned short *>(cls + tile.node_base + (int64_t)M * Lk)[lane];
        g = (int)(e & 63u);
        seg = (int)((e >> 6) & 63u);
    }
    const bool active = g < tile.G;
    const int d = tile.surf_base + (active ? g : 0);
    const int nn = active ? (meta[d] & 0xffff) : 0;
    const int64_t slot0 = first_slot[d];
    double *p = Tbuf + tile.node_base;
#pragma unroll
    for (int j = 0; j < M; j++) {
        const int i = seg * M + j;
        const int64_t idx = ((int64_t)(j >> 1) * Lk + lane) * 2 + (j & 1);
        if (to_state) {
            if (i < nn) state[slot0 + i] = p[idx];
        } else {
            p[idx] = (i < nn) ? state[slot0 + i] : 0.0;
        }
    }
}

__global__ void __launch_bounds__(256)
k_nodes_general(const GeneralTile *__restrict__ tiles, int n_tiles, double *__restrict__ Tbuf,
                const int32_t *__restrict__ meta, const int64_t *__restrict__ first_slot,
                double *__restrict__ state, int to_state) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= n_tiles) return;
    const GeneralTile tile = tiles[wave];
    const bool active = lane < tile.G;
    const int d = tile.surf_base + (active ? lane : 0);
    const int nn = active ? (meta[d] & 0xffff) : 0;
    const int64_t slot0 = first_slot[d];
    double *p = Tbuf + tile.node_base + lane;
    for (int i = 0; i < tile.n_max; i++) {
        if (to_state) {
            if (i < nn) state[slot0 + i] = p[(int64_t)i * kWave];
        } else {
            p[(int64_t)i * kWave] = (i < nn) ? state[slot0 + i] : 0.0;
        }
    }
}

// what: bit 0 inputs (solar, ir), bit 1 outputs (hs, flow)
__global__ void __launch_bounds__(256)
k_surf_scalars(int n_surf, SlotArrays sl, SideDyn *__restrict__ dyn, SideOut *__restrict__ out,
               const double *__restrict__ side_alpha, double *__restrict__ state, int to_state, int what) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_surf) return;
    const int S = n_surf;
    if (to_state) {
        if (what & 2) {
            const SideOut f = out[d], b = out[S + d];
            state[sl.hs_f[d]] = f.hs;
            state[sl.hs_b[d]] = b.hs;
            state[sl.flow_f[d]] = f.flow;
            state[sl.flow_b[d]] = b.flow;
        }
    } else {
        if (what & 1) {
            // solar clamps of ThermalSurfaceData::march (surface.rs:916-923): front NaN or < 0 -> 0;
            // back only NaN -> 0 (its second clause tests solar_front, already clamped)
            double sf = state[sl.solar_f[d]];
            if (sf != sf || sf < 0.0) sf = 0.0;
            double sb = state[sl.solar_b[d]];
            if (sb != sb) sb = 0.0;
            SideDyn f, b;
            f.solar = sf * side_alpha[d];      // fast classes: absorbed by the face node; others: factor 1
            f.rad_t = ir_to_rad_temperature(state[sl.ir_f[d]]);
            b.solar = sb * side_alpha[S + d];
            b.rad_t = ir_to_rad_temperature(state[sl.ir_b[d]]);
            dyn[d] = f;
            dyn[S + d] = b;
        }
        if (what & 2) {
            SideOut f, b;
            f.hs = state[sl.hs_f[d]];
            b.hs = state[sl.hs_b[d]];
            f.flow = state[sl.flow_f[d]];
            b.flow = state[sl.flow_b[d]];
            out[d] = f;
            out[S + d] = b;
        }
    }
}

__global__ void __launch_bounds__(256)
k_zone_scalars(int n_zones, const int64_t *__restrict__ zone_slot, double *__restrict__ zone_T,
               double *__restrict__ state, int to_state) {
    const int z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= n_zones) return;
    if (to_state) state[zone_slot[z]] = zone_T[z];
    else zone_T[z] = state[zone_slot[z]];
}

// heat_batch_march on a caller-owned state, compact transfers: what other modules write between two
// marches arrives as [solar_f[S] | solar_b[S] | ir_f[S] | ir_b[S] | zone T[Z]] in device surface order (gathered on
// the host, one pinned copy) — the clamps and conversions of k_surf_scalars, without the state mirror.
__global__ void __launch_bounds__(256)
k_inputs_compact(int n_surf, int n_zones, const double *__restrict__ in, const double *__restrict__ side_alpha,
                 SideDyn *__restrict__ dyn, double *__restrict__ zone_T, SlotArrays sl, double *__restrict__ mirror) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int S = n_surf;
    if (i < S) {
        if (mirror != nullptr) {
            // the state mirror keeps the caller's raw values of these slots: a download that copies whole runs of the
            // mirror into the caller's state (heat_batch_march) hands them back unchanged
            mirror[sl.solar_f[i]] = in[i];
            mirror[sl.solar_b[i]] = in[S + i];
            mirror[sl.ir_f[i]] = in[2 * (int64_t)S + i];
            mirror[sl.ir_b[i]] = in[3 * (int64_t)S + i];
        }
        double sf = in[i];
        if (sf != sf || sf < 0.0) sf = 0.0;  // surface.rs:916-923
        double sb = in[S + i];
        if (sb != sb) sb = 0.0;
        SideDyn f, b;
        f.solar = sf * side_alpha[i];
        f.rad_t = ir_to_rad_temperature(in[2 * (int64_t)S + i]);
        b.solar = sb * side_alpha[S + i];
        b.rad_t = ir_to_rad_temperature(in[3 * (int64_t)S + i]);
        dyn[i] = f;
        dyn[S + i] = b;
    }
    if (i < n_zones) zone_T[i] = in[4 * (int64_t)S + i];
}

// The outputs this path owns besides the node temperatures, compact: [hs_f, hs_b, flow_f, flow_b] per surface in
// the CALLER's surface order, then the zone temperatures.
__global__ void __launch_bounds__(256)
k_outputs_compact(int n_surf, int n_zones, const SideOut *__restrict__ out, const int32_t *__restrict__ orig_of,
                  const double *__restrict__ zone_T, double *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_surf) {
        const SideOut f = out[i], b = out[n_surf + i];
        double2 *p = reinterpret_cast<double2 *>(dst + 4 * (int64_t)orig_of[i]);
        p[0] = make_double2(f.hs, b.hs);
        p[1] = make_double2(f.flow, b.flow);
    }
    if (i < n_zones) dst[4 * (int64_t)n_surf + i] = zone_T[i];
}

__global__ void k_set_step(int *step_ptr, int v, int last) {
    step_ptr[0] = v;
    step_ptr[1] = last;
}

// The head of a march call in one launch: the call's weather and the zones' a0 / b0 terms from the pinned host staging
// buffers (read over PCIe by the kernel itself: three small copies on the stream cost 30-50 us of idle chip per call
// between them and the first kernel), and the sub-timestep counter.
__global__ void k_begin_march(const StepWeather *__restrict__ h_weather, StepWeather *__restrict__ weather, int n_sub,
                              const double *__restrict__ h_zone_ab, double *__restrict__ a0, double *__restrict__ b0,
                              int n_zones, int *step_ptr, int last) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sub) weather[i] = h_weather[i];
    if (i < n_zones) {
        a0[i] = h_zone_ab[i];
        b0[i] = h_zone_ab[n_zones + i];
    }
    if (i == 0) {
        step_ptr[0] = 0;
        step_ptr[1] = last;
    }
}

// ---------------------------------------------------------------------------
// Launch wrappers (host).
static inline int blocks_for_waves(int n_waves) { return (n_waves + 3) / 4; }
// dynamic LDS of the streamed palette kernels: the palettes of a 256-lane workgroup's tiles
static inline size_t pal_lds_bytes(const NodeArrays &na) { return (size_t)4 * kWave * na.pal_stride * sizeof(double); }

void launch_surfaces_fast(int M, int nm, int pal, int cav, double grid_share, const FastTile *tiles, int n_tiles,
                          const NodeArrays &na,
                          const SideArrays &sa, const StepWeather *weather, const int *step_ptr, int step_fixed,
                          const double *zone_T, int *flags, unsigned long long *nomass_iters, int n_cu, hipStream_t st, int reverse) {
    if (n_tiles <= 0) return;
    // persistent grid: as many 4-wave blocks as the chip holds at this kernel's occupancy (waves per SIMD by
    // VGPR count: M = 4 -> 5, M = 8 -> 3, M = 16 -> 2), capped by the number of tiles; n_cu: compute units of the
    // batch's device
    static const int tune = getenv("HEAT_AMD_PERSIST") ? atoi(getenv("HEAT_AMD_PERSIST")) : 1;
    const int blocks_per_cu = (M == 4 ? 5 : (M == 8 ? 3 : 2)) * tune;
    const int full_grid = blocks_for_waves(n_tiles);
    // A class that has the chip to itself runs persistently; classes that run side by side on several streams
    // launch one wave per tile and let the dispatcher interleave them (measured: dividing the chip between
    // persistent grids by work share was 1.7x slower on the ragged config).
    // (M = 4 tiles are too small for it: 1 M x 20 nodes ran 172 us persistent vs 155 us one wave per tile.)
    const bool persistent = tune > 0 && grid_share > 0.999 && M >= 8;
    const dim3 grid(persistent ? std::min(full_grid, n_cu * blocks_per_cu) : full_grid), block(256);
    FusedArgs no_fa{};
    no_fa.reverse = reverse;
#define HEAT_LAUNCH_FAST(MM, NN, PP, CC)                                                                          \
    hipLaunchKernelGGL((k_surfaces_fast<MM, NN, PP, CC, 0>), grid, block, PP ? pal_lds_bytes(na) : 0, st, tiles, n_tiles, na, sa, weather, \
                       step_ptr, step_fixed, zone_T, flags, nomass_iters, no_fa)
#define HEAT_LAUNCH_M(MM)                                     \
    switch ((nm ? 3 : 0) + (cav ? 2 : (pal ? 1 : 0))) {       \
    case 0: HEAT_LAUNCH_FAST(MM, 0, 0, 0); break;             \
    case 1: HEAT_LAUNCH_FAST(MM, 0, 1, 0); break;             \
    case 2: HEAT_LAUNCH_FAST(MM, 0, 1, 1); break;             \
    case 3: HEAT_LAUNCH_FAST(MM, 1, 0, 0); break;             \
    case 4: if (nm == 2) HEAT_LAUNCH_FAST(MM, 2, 1, 0); else HEAT_LAUNCH_FAST(MM, 1, 1, 0); break; \
    default: if (nm == 2) HEAT_LAUNCH_FAST(MM, 2, 1, 1); else HEAT_LAUNCH_FAST(MM, 1, 1, 1); break; \
    }
    if (M == 4) { HEAT_LAUNCH_M(4) } else if (M == 8) { HEAT_LAUNCH_M(8) } else { HEAT_LAUNCH_M(16) }
#undef HEAT_LAUNCH_M
#undef HEAT_LAUNCH_FAST
}

// grid: persistent — as many 4-wave blocks per compute unit as the variant's launch bounds admit — capped by the tile count.
void launch_surfaces_stream(int variant, const FastTile *tiles, int n_tiles, const NodeArrays &na, int64_t gen_base, const SideArrays &sa,
                            const StepWeather *weather, const int *step_ptr, int step_fixed, const double *zone_T,
                            int *flags, unsigned long long *nomass_iters, int n_cu, hipStream_t st, int reverse) {
    if (n_tiles <= 0) return;
    static const int per_cu_env = getenv("HEAT_AMD_STREAM_BLOCKS") ? atoi(getenv("HEAT_AMD_STREAM_BLOCKS")) : 0;
    const int per_cu = per_cu_env > 0 ? per_cu_env : (variant == kStreamLight ? 3 : 2);
    const int grid = std::min(blocks_for_waves(n_tiles), n_cu * per_cu);
#define HEAT_LAUNCH_STREAM(V)                                                                                       \
    hipLaunchKernelGGL(k_surfaces_stream<V>, dim3(grid), dim3(256), pal_lds_bytes(na), st, tiles, n_tiles, na, gen_base, sa, weather, \
                       step_ptr, step_fixed, zone_T, flags, nomass_iters, reverse)
    if (variant == kStreamWide) HEAT_LAUNCH_STREAM(kStreamWide);
    else if (variant == kStreamLight) HEAT_LAUNCH_STREAM(kStreamLight);
    else if (variant == kStreamChunks) HEAT_LAUNCH_STREAM(kStreamChunks);
    else HEAT_LAUNCH_STREAM(kStreamCav);
#undef HEAT_LAUNCH_STREAM
}

// Cluster-resident march of one class: one workgroup of `max_waves` (4 or 8) wavefronts per FusedBlock, fa.n_sub
// sub-timesteps in one launch. Palette classes without cavities only.
size_t fused_lds_bytes(int max_waves, int M, int pal_stride) {
    return (size_t)max_waves * kWave * ((pal_stride + (M == 16 ? M : 0)) * sizeof(double) + 2 * sizeof(double2)) +
           6 * kFusedMaxZones * sizeof(double) + (kFusedMaxZones + 2) * sizeof(int) + kFusedMaxEntries * sizeof(uint16_t);
}

template <int MM, int NN, int CC, int FW, int SM>
static hipError_t launch_fused_one(int grid_blocks, const FastTile *tiles, int n_tiles, const NodeArrays &na,
                                   const SideArrays &sa, const StepWeather *weather, int *flags,
                                   unsigned long long *nomass_iters, const FusedArgs &fa, hipStream_t st) {
    const size_t lds = fused_lds_bytes(FW, MM, na.pal_stride);
    // (per instantiation and device) dynamic LDS above the 64 KB default needs the attribute; it grows with the palette width
    constexpr int kMaxDevices = 64;
    static std::atomic<size_t> attr_bytes[kMaxDevices];  // 0: never set
    if (lds > 160 * 1024) return hipErrorInvalidValue;  // (the planner keeps wide-palette 16-node clusters to four waves)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = kMaxDevices - 1;
    if (lds > 64 * 1024 && lds > attr_bytes[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_surfaces_fast<MM, NN, 1, CC, FW, SM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev != kMaxDevices - 1) attr_bytes[dev].store(lds, std::memory_order_release);  // (an unknown device: set it every time)
    }
    hipLaunchKernelGGL((k_surfaces_fast<MM, NN, 1, CC, FW, SM>), dim3(grid_blocks), dim3(kWave * FW), lds, st, tiles,
                       n_tiles, na, sa, weather, nullptr, 0, nullptr, flags, nomass_iters, fa);
    return hipSuccess;
}

// mixed = 1: the workgroups also hold small-surface wavefronts; those run one universal variant per blocking factor
// (no-mass facings allowed, gas cavities allowed up to 8 nodes per lane). mixed = 2: teams (layout.hpp, FusedSuper):
// grid_blocks = n_teams * kTeamMax, fa.supers / team_zinfo / xbuf / tag_base set.
// How many workgroups of a fused variant one compute unit holds (two waves per SIMD by registers; three for the
// plain 4-node variants), also bounded by LDS.
int fused_blocks_per_cu(int M, int cav, int mixed, int max_waves, int pal_stride) {
    const int waves_per_simd = (M == 4 && !cav && !mixed) ? 3 : 2;
    const int by_regs = std::max(1, waves_per_simd * 4 / max_waves);
    const int by_lds = std::max<int>(1, (int)(160 * 1024 / fused_lds_bytes(max_waves, M, pal_stride)));
    return std::min(by_regs, by_lds);
}

// Workgroups of a TEAM variant one compute unit holds at once, as the runtime's occupancy arithmetic has it for the very
// kernel that will be launched with its dynamic LDS (the co-residency bound of a team launch: batch.hip).
template <int MM, int NN>
static int team_occupancy(size_t lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(&k_surfaces_fast<MM, NN, 1, 0, 4, 2>),
                                                     kWave * 4, lds) != hipSuccess)
        return 0;
    return n;
}
int fused_team_blocks_per_cu(int M, int nm, int pal_stride) {
    const size_t lds = fused_lds_bytes(4, M, pal_stride);
    if (M == 4) return nm ? team_occupancy<4, 1>(lds) : team_occupancy<4, 0>(lds);
    if (M == 8) return nm ? team_occupancy<8, 1>(lds) : team_occupancy<8, 0>(lds);
    return nm ? team_occupancy<16, 1>(lds) : team_occupancy<16, 0>(lds);
}

// grid_blocks workgroups for fa.n_blocks FusedBlocks: equal (fa.queue == nullptr) or fewer, with the work queue.
hipError_t launch_surfaces_fused(int M, int nm, int cav, int mixed, int max_waves, int grid_blocks, const FastTile *tiles,
                                 int n_tiles, const NodeArrays &na, const SideArrays &sa, const StepWeather *weather,
                                 int *flags, unsigned long long *nomass_iters, const FusedArgs &fa, hipStream_t st) {
    const int n_blocks = grid_blocks;
    if (n_blocks <= 0) return hipSuccess;
    if (cav && M == 16) return hipErrorInvalidValue;  // (the planner never asks for it)
#define HEAT_FUSED(MM, NN, CC, FW, SM) \
    launch_fused_one<MM, NN, CC, FW, SM>(n_blocks, tiles, n_tiles, na, sa, weather, flags, nomass_iters, fa, st)
    if (mixed == 2) {  // teams of workgroups (clusters larger than one): four wavefronts each, no cavities, no small surfaces
        if (cav || max_waves > 4) return hipErrorInvalidValue;
        if (M == 4) return nm ? HEAT_FUSED(4, 1, 0, 4, 2) : HEAT_FUSED(4, 0, 0, 4, 2);
        if (M == 8) return nm ? HEAT_FUSED(8, 1, 0, 4, 2) : HEAT_FUSED(8, 0, 0, 4, 2);
        return nm ? HEAT_FUSED(16, 1, 0, 4, 2) : HEAT_FUSED(16, 0, 0, 4, 2);
    }
    if (mixed) {
        if (M == 4) return max_waves <= 4 ? HEAT_FUSED(4, 1, 1, 4, 1) : HEAT_FUSED(4, 1, 1, 8, 1);
        if (M == 8) return max_waves <= 4 ? HEAT_FUSED(8, 1, 1, 4, 1) : HEAT_FUSED(8, 1, 1, 8, 1);
        return max_waves <= 4 ? HEAT_FUSED(16, 1, 0, 4, 1) : HEAT_FUSED(16, 1, 0, 8, 1);
    }
    if (nm == 2 && !cav && M <= 8) {  // walls with no-mass chunks other than facings (8 / 4 nodes per lane: the registers)
        if (M == 4) return max_waves <= 4 ? HEAT_FUSED(4, 2, 0, 4, 0) : HEAT_FUSED(4, 2, 0, 8, 0);
        return max_waves <= 4 ? HEAT_FUSED(8, 2, 0, 4, 0) : HEAT_FUSED(8, 2, 0, 8, 0);
    }
#define HEAT_FUSED_NW(MM, CC)                                                    \
    (max_waves <= 4 ? (nm ? HEAT_FUSED(MM, 1, CC, 4, 0) : HEAT_FUSED(MM, 0, CC, 4, 0)) \
                    : (nm ? HEAT_FUSED(MM, 1, CC, 8, 0) : HEAT_FUSED(MM, 0, CC, 8, 0)))
    if (M == 4) return cav ? HEAT_FUSED_NW(4, 1) : HEAT_FUSED_NW(4, 0);
    if (M == 8) return cav ? HEAT_FUSED_NW(8, 1) : HEAT_FUSED_NW(8, 0);
    return HEAT_FUSED_NW(16, 0);
#undef HEAT_FUSED_NW
#undef HEAT_FUSED
}

void launch_surfaces_general(const GeneralTile *tiles, int n_tiles, const NodeArrays &na, int64_t gen_base,
                             const SideArrays &sa, const CavityDev *cavs, double *scratch,
                             const StepWeather *weather, const int *step_ptr, int step_fixed,
                             const double *zone_T, int *flags, unsigned long long *nomass_iters, hipStream_t st) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(k_surfaces_general, dim3(blocks_for_waves(n_tiles)), dim3(256), 0, st, tiles, n_tiles, na,
                       gen_base, sa, cavs, scratch, weather, step_ptr, step_fixed, zone_T, flags, nomass_iters);
}

void launch_surfaces_small(int with_cavities, const GeneralTile *tiles, int n_tiles, const NodeArrays &na,
                           int64_t gen_base, const SideArrays &sa, const CavityDev *cavs,
                           const StepWeather *weather, const int *step_ptr, int step_fixed, const double *zone_T,
                           int *flags, unsigned long long *nomass_iters, hipStream_t st) {
    if (n_tiles <= 0) return;
    if (with_cavities)
        hipLaunchKernelGGL(k_surfaces_small<1>, dim3(blocks_for_waves(n_tiles)), dim3(256), 0, st, tiles, n_tiles, na,
                           gen_base, sa, cavs, weather, step_ptr, step_fixed, zone_T, flags, nomass_iters);
    else
        hipLaunchKernelGGL(k_surfaces_small<0>, dim3(blocks_for_waves(n_tiles)), dim3(256), 0, st, tiles, n_tiles, na,
                           gen_base, sa, cavs, weather, step_ptr, step_fixed, zone_T, flags, nomass_iters);
}

void launch_zones(const int64_t *zone_off, const ZoneEntry *entries, const ZoneContrib *zc,
                  const double *a0, const double *b0, const double *zone_vol, double *zone_T, double *partial,
                  int n_zones, double dt, int *step_ptr, int *flags, int mode, const int32_t *zlist, int n_list,
                  const int32_t *slot_of, int n_shared, int rows, hipStream_t st) {
    const int n_z = (mode == 2 || mode == 3) ? n_list : n_zones;
    if (mode == 2 && n_z <= 0) return;
    // rows: a zone per row of 16 lanes (the batch's zones have few walls each), else a zone per wavefront
    if (rows) {
        const int nb = n_z > 0 ? (n_z + 15) / 16 : 1;
        hipLaunchKernelGGL(k_zones<1>, dim3(nb), dim3(256), 0, st, zone_off, entries, zc, a0, b0, zone_vol, zone_T,
                           partial, n_zones, dt, step_ptr, flags, mode, zlist, n_list, slot_of, n_shared);
    } else {
        const int nb = n_z > 0 ? blocks_for_waves(n_z) : 1;
        hipLaunchKernelGGL(k_zones<0>, dim3(nb), dim3(256), 0, st, zone_off, entries, zc, a0, b0, zone_vol, zone_T,
                           partial, n_zones, dt, step_ptr, flags, mode, zlist, n_list, slot_of, n_shared);
    }
}

void launch_zone_update_shared(const double *gathered, int n_blocks, const int32_t *shared_zone, int n_shared,
                               const double *a0, const double *b0, const double *zone_vol, double *zone_T,
                               double dt, int *flags, hipStream_t st) {
    if (n_shared <= 0) return;
    hipLaunchKernelGGL(k_zone_update_shared, dim3((n_shared + 255) / 256), dim3(256), 0, st, gathered, n_blocks,
                       shared_zone, n_shared, a0, b0, zone_vol, zone_T, dt, flags);
}

void launch_zone_update(const double *gathered, int n_blocks, const double *a0, const double *b0,
                        const double *zone_vol, double *zone_T, int n_zones, double dt, int *step_ptr,
                        int *flags, hipStream_t st) {
    const int nb = n_zones > 0 ? (n_zones + 255) / 256 : 1;
    hipLaunchKernelGGL(k_zone_update, dim3(nb), dim3(256), 0, st, gathered, n_blocks, a0, b0, zone_vol, zone_T,
                       n_zones, dt, step_ptr, flags);
}

void launch_nodes_fast(int M, const FastTile *tiles, int n_tiles, double *Tbuf, const int32_t *meta,
                       const int64_t *first_slot, double *state, int to_state, const uint8_t *cls, hipStream_t st) {
    if (n_tiles <= 0) return;
    const dim3 grid(blocks_for_waves(n_tiles)), block(256);
    switch (M) {
    case 4: hipLaunchKernelGGL(k_nodes_fast<4>, grid, block, 0, st, tiles, n_tiles, Tbuf, meta, first_slot, state, to_state, cls); break;
    case 8: hipLaunchKernelGGL(k_nodes_fast<8>, grid, block, 0, st, tiles, n_tiles, Tbuf, meta, first_slot, state, to_state, cls); break;
    default: hipLaunchKernelGGL(k_nodes_fast<16>, grid, block, 0, st, tiles, n_tiles, Tbuf, meta, first_slot, state, to_state, cls); break;
    }
}

void launch_nodes_general(const GeneralTile *tiles, int n_tiles, double *Tbuf, const int32_t *meta,
                          const int64_t *first_slot, double *state, int to_state, hipStream_t st) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(k_nodes_general, dim3(blocks_for_waves(n_tiles)), dim3(256), 0, st, tiles, n_tiles, Tbuf,
                       meta, first_slot, state, to_state);
}

void launch_surf_scalars(int n_surf, const SlotArrays &sl, SideDyn *dyn, SideOut *out, const double *side_alpha,
                         double *state, int to_state, int what, hipStream_t st) {
    if (n_surf <= 0) return;
    hipLaunchKernelGGL(k_surf_scalars, dim3((n_surf + 255) / 256), dim3(256), 0, st, n_surf, sl, dyn, out, side_alpha,
                       state, to_state, what);
}

void launch_zone_scalars(int n_zones, const int64_t *zone_slot, double *zone_T, double *state, int to_state,
                         hipStream_t st) {
    if (n_zones <= 0) return;
    hipLaunchKernelGGL(k_zone_scalars, dim3((n_zones + 255) / 256), dim3(256), 0, st, n_zones, zone_slot, zone_T,
                       state, to_state);
}

void launch_inputs_compact(int n_surf, int n_zones, const double *in, const double *side_alpha, SideDyn *dyn, double *zone_T,
                           const SlotArrays &sl, double *mirror, hipStream_t st) {
    const int n = std::max(n_surf, n_zones);
    if (n <= 0) return;
    hipLaunchKernelGGL(k_inputs_compact, dim3((n + 255) / 256), dim3(256), 0, st, n_surf, n_zones, in, side_alpha, dyn, zone_T,
                       sl, mirror);
}

void launch_outputs_compact(int n_surf, int n_zones, const SideOut *out, const int32_t *orig_of, const double *zone_T,
                            double *dst, hipStream_t st) {
    const int n = std::max(n_surf, n_zones);
    if (n <= 0) return;
    hipLaunchKernelGGL(k_outputs_compact, dim3((n + 255) / 256), dim3(256), 0, st, n_surf, n_zones, out, orig_of, zone_T, dst);
}

void launch_begin_march(const StepWeather *h_weather, StepWeather *weather, int n_sub, const double *h_zone_ab, double *a0,
                        double *b0, int n_zones, int *step_ptr, hipStream_t st) {
    const int n = std::max(std::max(n_sub, n_zones), 1);
    hipLaunchKernelGGL(k_begin_march, dim3((n + 255) / 256), dim3(256), 0, st, h_weather, weather, n_sub, h_zone_ab, a0, b0,
                       n_zones, step_ptr, n_sub - 1);
}

void launch_set_step(int *step_ptr, int v, int last, hipStream_t st) {
    hipLaunchKernelGGL(k_set_step, dim3(1), dim3(1), 0, st, step_ptr, v, last);
}

}  // namespace heat

#if defined(HEAT_STAMPS) || defined(HEAT_STREAM_STAMPS)
extern "C" int heat_debug_stamps(unsigned long long *dst, int n_blocks) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(heat::g_stamps), (size_t)n_blocks * heat::kStampsPerBlock * sizeof(unsigned long long));
}
#endif
