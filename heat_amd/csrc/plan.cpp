// plan.cpp — host-only planner of a heat batch (plan.hpp). Compiled into libheat_amd.so by hipcc and, on its own, by
// g++ for the sanitizer tests: no HIP here.
#include "plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace heat {

const int kFastM[kNumFast] = {4, 4, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8, 16, 16, 16, 16, 16, 16};
const int kFastNM[kNumFast] = {0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1};
const int kFastPAL[kNumFast] = {0, 1, 1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 0, 1, 1};
const int kFastCAV[kNumFast] = {0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};

namespace {

int failp(std::string &err, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
}

struct Placed {
    int64_t s;   // original surface index
    int n;       // node count
    int cls;     // 0..kNumFast-1 fast classes, kGeneral = catch-all
    int k;       // lanes per surface (fast)
    int blk;     // cluster-resident march: workgroup number, -1 = streamed
};

// Tiles of a cluster-resident workgroup: its surfaces are packed into wavefronts whatever their lane counts (a
// mixed tile carries a lane table, layout.hpp), in ascending order of k — what the tiling below does.
int tiles_needed_packed(const int (&cnt)[kWave + 1]) {
    int tiles = 0, lanes = 0;
    for (int k = 1; k <= kWave; k++)
        for (int q = 0; q < cnt[k]; q++) {
            if (lanes + k > kWave) { tiles++; lanes = 0; }
            lanes += k;
        }
    return tiles + (lanes > 0);
}

// What kind of kernel a surface needs (before the blocking factor M is chosen).
struct Category {
    int kind;  // kSmall, kSmallCav, kGeneral, or 0 = fast path
    int nm;    // fast: has no-mass chunks (one or two nodes each, between massive nodes and / or a face)
    int ncav;  // fast: gas cavities between massive nodes
    int pal;   // fast: the per-node constants fit a palette
    int m_ok;  // fast: blocking factors whose lanes hold every chunk whole (bit 0: 4, bit 1: 8, bit 2: 16 nodes per lane)
    int chunky = 0;  // fast: has chunks other than one-node facings (inside the wall, or of two nodes)
    int wide = 0;    // fast, palette form: needs the wide palette (more than 8 V or 4 U entries, with the 0.0 of entry 0)
    int tiny = 0;    // fast, palette form: fits the tiny palette (4 V and 4 U entries at most)
};
inline int m_bit(int M) { return M == 4 ? 1 : (M == 8 ? 2 : 4); }

// Decides whether a surface can take the register-resident fast path:
// solid conductances only, solar absorbed at the two faces only, every interior node massive; the
// face nodes may be no-mass facings (each then is an isolated one-node no-mass chunk).
Category categorize(const heat_batch_desc *d, int64_t s, int n, const heat_batch_options &opt) {
    Category r{kGeneral, 0, 0, 0, 7};
    if (opt.force_general) return r;
    const int64_t o = d->node_offset[s];
    if (n <= 4) {
        bool all_nomass = true;
        for (int i = 0; i < n; i++) all_nomass = all_nomass && (d->mass[o + i] < kMassThreshold);
        bool has_cav = false;
        for (int i = 0; i < n; i++)
            has_cav = has_cav || (d->seg_cavity && d->n_cavities > 0 && d->seg_cavity[o + i] >= 0);
        if (all_nomass) {
            r.kind = has_cav ? kSmallCav : kSmall;
            return r;
        }
    }
    if (n < 2) return r;
    int nm = 0, ncav = 0, m_ok = 7;
    bool facings_only = true;
    auto is_cav = [&](int i) { return d->seg_cavity && d->n_cavities > 0 && d->seg_cavity[o + i] >= 0; };
    for (int i = 0; i < n; i++) {
        if (is_cav(i)) {
            // a cavity on the fast path sits between two massive nodes (its conductance is then needed
            // once per sub-timestep, not once per pass of a no-mass loop)
            if (i + 1 >= n || d->mass[o + i] < kMassThreshold || d->mass[o + i + 1] < kMassThreshold) return r;
            if (++ncav > 2) return r;
        }
        if (i > 0 && d->front_alpha[o + i] != 0.0) return r;           // solar absorbed inside
        if (i < n - 1 && d->back_alpha[o + i] != 0.0) return r;
    }
    // No-mass chunks (get_chunks, discretization.rs:144-160): one or two nodes each — a thin facing, two light
    // layers at a face, two light layers inside a cavity wall; longer runs go to the catch-all kernel. A two-node
    // chunk has to sit in one lane, and a lane solves two chunks at most.
    for (int i = 0; i < n;) {
        if (d->mass[o + i] >= kMassThreshold) { i++; continue; }
        int e = i;
        while (e < n && d->mass[o + e] < kMassThreshold) e++;
        if (e - i > 2) return r;
        nm = 1;
        if (!(e - i == 1 && (i == 0 || i == n - 1))) facings_only = false;
        for (int M : {4, 8, 16})
            if (e - i == 2 && i / M != (i + 1) / M) m_ok &= ~m_bit(M);
        i = e;
    }
    for (int M : {4, 8, 16}) {
        int in_lane = 0, lane = -1;
        for (int i = 0; i < n; i++) {
            const bool start = d->mass[o + i] < kMassThreshold && (i == 0 || d->mass[o + i - 1] >= kMassThreshold);
            if (i / M != lane) { lane = i / M; in_lane = 0; }
            if (start && ++in_lane > 2) m_ok &= ~m_bit(M);
        }
    }
    if (m_ok == 0) return r;
    // Palette form when the wall has few distinct constants (entry 0 of each palette is 0.0).
    int pal = opt.no_palette ? 0 : 1;
    int nu = 1, nv = 1;
    if (pal) {
        double vv[kPalV], uu[kPalU];
        vv[0] = 0.0;
        uu[0] = 0.0;
        for (int i = 0; i < n && pal; i++) {
            const double mass = d->mass[o + i];
            const double v = (mass >= kMassThreshold) ? d->dt / mass : 0.0;
            const double u = is_cav(i) ? 0.0 : d->uvalue[o + i];
            int f = -1;
            for (int q = 0; q < nv; q++) if (vv[q] == v) f = q;
            if (f < 0) { if (nv == kPalVMark1) pal = 0; else vv[nv++] = v; }  // (indices 14, 15 are chunk marks)
            f = -1;
            for (int q = 0; q < nu; q++) if (uu[q] == u) f = q;
            if (f < 0) { if (nu == kPalU) pal = 0; else uu[nu++] = u; }
        }
    }
    if (ncav > 0 && !pal) return r;  // the cavity variant exists in palette form only
    if (nm && !facings_only && !pal) return r;  // chunks other than one-node facings are marked in the class bytes
    r.kind = 0;
    r.nm = nm;
    r.ncav = ncav;
    r.pal = pal;
    r.m_ok = m_ok;
    r.chunky = (nm && !facings_only) ? 1 : 0;
    r.wide = (pal && (nv > kPalVNarrow || nu > kPalUNarrow)) ? 1 : 0;
    r.tiny = (pal && nv <= kPalVTiny && nu <= kPalUTiny) ? 1 : 0;
    return r;
}

// Nodes are padded to a multiple of M inside the last lane. Measured cost per padded node
// (1 M x 32 and 1 M x 20 nodes, profiles/README.md): M = 16 : 8 : 4 = 1.00 : 1.03 : 1.30 — larger blocks
// amortise the per-surface boundary work over more nodes.
double padded_cost(int n, int M) {
    const double w = (M == 4) ? 1.30 : (M == 8 ? 1.03 : 1.00);
    return (double)((n + M - 1) / M * M) * w;
}

// nm: the wall has a no-mass facing. Its 16-node variant (k_surfaces_fast<16, 1, 1, 0>) needs 266 registers — one
// wavefront per SIMD, 3.1 TB/s against 6.2 for the all-massive walls (profiles/README.md, round 2) — so walls with
// facings take 8 nodes per lane at most.
int choose_M(int n, int nm, int m_ok, const heat_batch_options &opt) {
    if (opt.nodes_per_lane != 0) return (m_ok & m_bit(opt.nodes_per_lane)) ? opt.nodes_per_lane : 0;
    int M = 0;
    for (int m : {4, 8, 16}) {
        if ((nm && m == 16 && (m_ok & 3)) || !(m_ok & m_bit(m))) continue;
        if (M == 0 || padded_cost(n, m) < padded_cost(n, M)) M = m;
    }
    return M;  // 0: no blocking factor holds the wall's chunks -> catch-all
}

int fast_class(int M, const Category &c) { return (M == 4 ? 0 : (M == 8 ? 6 : 12)) + c.nm * 3 + (c.ncav > 0 ? 2 : c.pal); }


}  // namespace

int check_desc(const heat_batch_desc *d, std::string &err) {
    if (!d) return failp(err, HEAT_E_INVALID_ARG, "descriptor is NULL");
    if (d->abi_version != HEAT_AMD_ABI_VERSION)
        return failp(err, HEAT_E_INVALID_ARG, "abi_version %d, library is %d", d->abi_version, HEAT_AMD_ABI_VERSION);
    if (d->n_surfaces < 0 || d->n_zones < 0 || d->n_cavities < 0 || d->n_state < 0)
        return failp(err, HEAT_E_INVALID_ARG, "negative count in descriptor");
    if (!(d->dt > 0.0)) return failp(err, HEAT_E_INVALID_ARG, "dt must be positive");
    const void *need[] = {d->node_offset, d->mass, d->uvalue, d->front_alpha, d->back_alpha, d->front_kind,
                          d->back_kind, d->front_zone, d->back_zone, d->front_ambient, d->back_ambient,
                          d->front_emissivity, d->back_emissivity, d->area, d->perimeter, d->cos_tilt,
                          d->normal_x, d->normal_y, d->wind_modifier, d->first_node_slot, d->hs_front_slot,
                          d->hs_back_slot, d->flow_front_slot, d->flow_back_slot, d->solar_front_slot,
                          d->solar_back_slot, d->ir_front_slot, d->ir_back_slot};
    if (d->n_surfaces > 0)
        for (const void *p : need)
            if (!p) return failp(err, HEAT_E_INVALID_ARG, "a required per-surface array is NULL");
    if (d->n_zones > 0 && (!d->zone_volume || !d->zone_slot))
        return failp(err, HEAT_E_INVALID_ARG, "zone arrays are NULL");
    if (d->n_cavities > 0 && (!d->cavities || !d->seg_cavity))
        return failp(err, HEAT_E_INVALID_ARG, "cavity arrays are NULL");
    if ((d->front_hs_fix == nullptr) != (d->back_hs_fix == nullptr))
        return failp(err, HEAT_E_INVALID_ARG, "front_hs_fix and back_hs_fix must both be given or both be NULL");
    const int64_t ns = d->n_state;
    auto slot_ok = [&](int64_t v) { return v >= 0 && v < ns; };
    for (int64_t z = 0; z < d->n_zones; z++)
        if (!slot_ok(d->zone_slot[z])) return failp(err, HEAT_E_SIZE, "zone %lld: slot out of range", (long long)z);
    if (d->n_surfaces > 0 && d->node_offset[0] != 0) return failp(err, HEAT_E_INVALID_ARG, "node_offset[0] != 0");
    for (int64_t s = 0; s < d->n_surfaces; s++) {
        const int64_t n = d->node_offset[s + 1] - d->node_offset[s];
        if (n < 1) return failp(err, HEAT_E_INVALID_ARG, "surface %lld has %lld nodes", (long long)s, (long long)n);
        if (n > kMaxNodesGeneral)
            return failp(err, HEAT_E_TOO_MANY_NODES, "surface %lld has %lld nodes (max %d)", (long long)s, (long long)n,
                        kMaxNodesGeneral);
        const int fk = d->front_kind[s], bk = d->back_kind[s];
        if (fk == HEAT_BOUNDARY_GROUND || bk == HEAT_BOUNDARY_GROUND)
            return failp(err, HEAT_E_GROUND_BOUNDARY, "surface %lld: Ground boundary is not supported (reference panics)", (long long)s);
        if (fk < 0 || fk > 2 || bk < 0 || bk > 2)
            return failp(err, HEAT_E_INVALID_ARG, "surface %lld: unknown boundary kind", (long long)s);
        if (fk == HEAT_BOUNDARY_SPACE && (d->front_zone[s] < 0 || d->front_zone[s] >= d->n_zones))
            return failp(err, HEAT_E_SIZE, "surface %lld: front zone out of range", (long long)s);
        if (bk == HEAT_BOUNDARY_SPACE && (d->back_zone[s] < 0 || d->back_zone[s] >= d->n_zones))
            return failp(err, HEAT_E_SIZE, "surface %lld: back zone out of range", (long long)s);
        if (d->first_node_slot[s] < 0 || d->first_node_slot[s] + n > ns)
            return failp(err, HEAT_E_SIZE, "surface %lld: node slots out of range", (long long)s);
        if (!slot_ok(d->hs_front_slot[s]) || !slot_ok(d->hs_back_slot[s]) || !slot_ok(d->flow_front_slot[s]) ||
            !slot_ok(d->flow_back_slot[s]) || !slot_ok(d->solar_front_slot[s]) || !slot_ok(d->solar_back_slot[s]) ||
            !slot_ok(d->ir_front_slot[s]) || !slot_ok(d->ir_back_slot[s]))
            return failp(err, HEAT_E_SIZE, "surface %lld: scalar slot out of range", (long long)s);
        const int64_t o = d->node_offset[s];
        for (int64_t i = 0; i < n; i++) {
            const bool cav = d->seg_cavity && d->n_cavities > 0 && d->seg_cavity[o + i] >= 0;
            if (cav && d->seg_cavity[o + i] >= d->n_cavities)
                return failp(err, HEAT_E_SIZE, "surface %lld: cavity index out of range", (long long)s);
            if (!cav && std::isnan(d->uvalue[o + i]))
                return failp(err, HEAT_E_UVALUE_NONE, "surface %lld node %lld: UValue::None", (long long)s, (long long)i);
            // (a node is massive iff mass >= 1e-5, discretization.rs:149,155: a NaN is neither massive nor no-mass — the
            // reference's chunks would silently leave the node out; here it is refused. Found by tools/fuzz_desc.py as an
            // endless loop of the planner.)
            if (std::isnan(d->mass[o + i]))
                return failp(err, HEAT_E_INVALID_ARG, "surface %lld node %lld: thermal mass is NaN", (long long)s, (long long)i);
        }
    }
    return HEAT_OK;
}

int make_plan(const heat_batch_desc *d, const heat_batch_options &opt, Plan &p, std::string &err) {
    int rc = check_desc(d, err);
    if (rc) return rc;
    // (heat_batch_create_ex refuses these before it comes here; heat_plan_check comes straight)
    if (opt.nodes_per_lane != 0 && opt.nodes_per_lane != 4 && opt.nodes_per_lane != 8 && opt.nodes_per_lane != 16)
        return failp(err, HEAT_E_INVALID_ARG, "nodes_per_lane must be 0, 4, 8 or 16");
    if (opt.n_ranks > 1 && (opt.rank < 0 || opt.rank >= opt.n_ranks))
        return failp(err, HEAT_E_INVALID_ARG, "rank %d outside [0, %d)", opt.rank, opt.n_ranks);
    const int64_t S = d->n_surfaces, Z = d->n_zones;
    p = Plan();
    p.n_surf = S;
    p.n_zones = Z;
    p.n_state = d->n_state;
    p.n_cav = d->n_cavities;
    p.dt = d->dt;
    p.n_nodes = S > 0 ? d->node_offset[S] : 0;
    p.algorithmic_bytes = 32 * p.n_nodes + 152 * S;  // 32 n + 152 bytes per surface per sub-timestep

    // ---- classify ----
    std::vector<Placed> placed(S);
    std::vector<Category> cat(S);
    for (int64_t s = 0; s < S; s++) {
        const int n = (int)(d->node_offset[s + 1] - d->node_offset[s]);
        cat[s] = categorize(d, s, n, opt);
        int cls = cat[s].kind, k = 1;
        if (cat[s].kind == 0) {
            const int M = choose_M(n, cat[s].nm, cat[s].m_ok, opt);
            k = M ? (n + M - 1) / M : kWave + 1;
            cls = (k > kWave) ? kGeneral : fast_class(M, cat[s]);
            if (k > kWave) { cat[s].kind = kGeneral; k = 1; }
        }
        placed[s] = Placed{s, n, cls, k, -1};
    }
    if (opt.nodes_per_lane == 0) {
        // A class of few surfaces is a launch of its own that cannot fill the chip (1 M ragged walls: the two 4-node
        // classes, 6 % of the surfaces, took 10 us each — as long as classes ten times their size). A blocking
        // factor that holds less than a twelfth of the fast-path nodes hands its walls to the 8-node classes.
        double nodes_by_M[3] = {0, 0, 0}, all_nodes = 0;
        for (int64_t s = 0; s < S; s++)
            if (placed[s].cls < kNumFast) {
                nodes_by_M[placed[s].cls / 6] += placed[s].n;
                all_nodes += placed[s].n;
            }
        for (int mi : {0, 2}) {
            if (nodes_by_M[mi] == 0 || nodes_by_M[mi] * 12 >= all_nodes || nodes_by_M[1] == 0) continue;
            for (int64_t s = 0; s < S; s++)
                if (placed[s].cls < kNumFast && placed[s].cls / 6 == mi && (placed[s].n + 7) / 8 <= kWave && (cat[s].m_ok & m_bit(8))) {
                    placed[s].k = (placed[s].n + 7) / 8;
                    placed[s].cls = fast_class(8, cat[s]);
                }
        }
    }

    // ---- cluster-resident march: zone-connected clusters -> workgroups (layout.hpp, FusedBlock) ----
    // A cluster is a connected component of the graph "zone - surface facing it"; its surfaces exchange heat
    // only through its own zones (model.rs:556-590), so a workgroup that holds all of them can march any number
    // of sub-timesteps without leaving the chip. A cluster is fused when every surface of it is a palette-form
    // fast-path wall without cavities and it fits kFusedMaxWaves tiles / kFusedMaxZones zones; its surfaces then
    // share one blocking factor (4 or 8: the 16-node variant does not fit the register file with the state that
    // lives across sub-timesteps). Surfaces that face no zone at all are clusters of one and are packed freely.
    // mixed: holds small-surface tiles too. super >= 0: member `member` of a team (layout.hpp, FusedSuper); zinfo: per
    // zone of the block its exchange slot | members << 16
    struct BlockPlan { int cls; bool mixed; std::vector<int32_t> zones; int super = -1; int member = 0; std::vector<uint32_t> zinfo; };
    int n_supers = 0;
    std::vector<BlockPlan> blocks;
    // no_fusion: 0 = fuse the clusters the cost model below expects to gain, 1 = never, 2 = every cluster that can be
    const bool fuse = opt.no_fusion != 1 && !opt.force_general && !opt.no_palette;
    const bool fuse_always = opt.no_fusion == 2;
    const std::vector<Placed> placed_streamed = placed;  // the plan without any cluster-resident march
    if (fuse && S > 0) {
        auto is_small = [&](int64_t s) { return cat[s].kind == kSmall || cat[s].kind == kSmallCav; };
        // (walls with no-mass chunks other than one-node facings: with 8 or 4 nodes per lane, not beside small surfaces or
        // gas cavities — the variants that carry the chunk loop, kernels.hip)
        auto fusable = [&](int64_t s) { return (cat[s].kind == 0 && cat[s].pal) || is_small(s); };
        auto zone_of_side = [&](int64_t s, int side) -> int32_t {
            const int kind = side ? d->back_kind[s] : d->front_kind[s];
            return kind == HEAT_BOUNDARY_SPACE ? (side ? d->back_zone[s] : d->front_zone[s]) : -1;
        };
        std::vector<int32_t> uf(Z);
        std::iota(uf.begin(), uf.end(), 0);
        auto find = [&](int32_t x) {
            while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; }
            return x;
        };
        for (int64_t s = 0; s < S; s++) {
            const int32_t zf = zone_of_side(s, 0), zb = zone_of_side(s, 1);
            if (zf >= 0 && zb >= 0) {
                const int32_t a = find(zf), c = find(zb);
                if (a != c) uf[std::max(a, c)] = std::min(a, c);
            }
        }
        // per cluster (root zone): its surfaces (CSR), whether all of them are fusable
        std::vector<int64_t> coff(Z + 1, 0);
        std::vector<uint8_t> cok(Z, 1);
        auto root_of = [&](int64_t s) -> int32_t {
            const int32_t zf = zone_of_side(s, 0), zb = zone_of_side(s, 1);
            return zf >= 0 ? find(zf) : (zb >= 0 ? find(zb) : -1);
        };
        std::vector<int32_t> sroot(S);
        for (int64_t s = 0; s < S; s++) {
            sroot[s] = root_of(s);
            if (sroot[s] >= 0) {
                coff[sroot[s] + 1]++;
                if (!fusable(s)) cok[sroot[s]] = 0;
            }
        }
        for (int64_t z = 0; z < Z; z++) coff[z + 1] += coff[z];
        std::vector<int64_t> csurf(Z > 0 ? coff[Z] : 0), ccur(coff.begin(), coff.end() - 1);
        for (int64_t s = 0; s < S; s++)
            if (sroot[s] >= 0) csurf[ccur[sroot[s]]++] = s;
        std::vector<std::vector<int32_t>> czones(Z);  // zones of each root
        for (int64_t z = 0; z < Z; z++) czones[find((int32_t)z)].push_back((int32_t)z);

        // open workgroup per class: surfaces per k, zones so far
        struct Open { int blk = -1; int cnt[kWave + 1] = {}; int nsmall = 0; int nz = 0; int ne = 0; };
        Open open[2 * kNumFast];  // [class][mixed]
        auto new_block = [&](int cls, bool mixed) {
            blocks.push_back(BlockPlan{cls, mixed, {}, -1, 0, {}});
            return (int)blocks.size() - 1;
        };
        auto small_tiles = [](int n_small) { return (n_small + kWave - 1) / kWave; };
        // Cost model (measured on MI355X, profiles/README.md): what the cluster-resident march of a cluster costs by
        // blocking factor and tile count (below) against what streaming it costs — its algorithmic bytes at the
        // ~5.5 TB/s the streamed kernels sustain, plus k_zones' share. The cheapest blocking factor is taken.
        constexpr double kStreamBytesPerNs = 5500.0, kZoneNs = 1.8;
        // ns per cluster and sub-timestep. Up to four tiles (two or three workgroups per compute unit): per tile —
        // 16 nodes per lane 2.05 (1 M x 32: 82 us / 40 000 tiles), 8: 1.8 (1 M x 13: 72 us), 4: 1.15 (1 M x 8: 46 us).
        // Five to eight tiles (a workgroup has a compute unit to itself, however many of its wavefronts work): per
        // workgroup — 16: 16.5 (1 M x 48: 165 us / 10 000), 8: 14.8 (1 M x 32: 148 us), 4: 12 (1 M x 10: 118 us).
        auto cluster_ns = [](int m, int tiles) {
            if (tiles <= 4) return tiles * (m == 16 ? 2.05 : (m == 8 ? 1.8 : 1.15));
            return m == 16 ? 16.5 : (m == 8 ? 14.8 : 12.0);
        };
        auto tile_ns = [](int m) { return m == 16 ? 2.05 : (m == 8 ? 1.8 : 1.15); };  // (surfaces that face no zone)
        auto fused_cost = [&](int n, int m) {  // (explicit nodes_per_lane, lone surfaces: lanes of the surface)
            return (double)((n + m - 1) / m);
        };
        const int ms_all[3] = {4, 8, 16};
        // LDS: with wide palettes (layout.hpp) a workgroup of eight 16-node wavefronts would need 180 KB
        bool wide_batch = false;
        for (int64_t s = 0; s < S; s++) wide_batch = wide_batch || (cat[s].kind == 0 && cat[s].wide);
        auto max_tiles = [&](int m) { return (m == 16 && wide_batch) ? 4 : kFusedMaxWaves; };
        for (int64_t r = 0; r < Z; r++) {
            if (find((int32_t)r) != r || !cok[r] || coff[r + 1] == coff[r]) continue;
            // one blocking factor for the cluster: the cheapest that keeps every surface at two lanes or more
            // (gas cavities: 4 or 8 nodes per lane only — the 16-node cavity variant does not fit the registers)
            // Small all-no-mass surfaces (glazing, thin walls) of the cluster get wavefronts of their own in the
            // workgroup (one lane per surface); such a "mixed" workgroup runs the universal kernel variant of its
            // blocking factor: no-mass facings allowed, gas cavities allowed (up to 8 nodes per lane).
            bool any_cav = false, any_chunky = false;
            int n_small = 0;
            for (int64_t q = coff[r]; q < coff[r + 1]; q++) {
                if (is_small(csurf[q])) n_small++;
                else {
                    any_cav = any_cav || cat[csurf[q]].ncav > 0;
                    any_chunky = any_chunky || cat[csurf[q]].chunky != 0;
                }
            }
            const bool mixed = n_small > 0;
            int M = opt.nodes_per_lane;
            if (M == 16 && (any_cav || any_chunky)) continue;  // streamed
            if (any_chunky && (mixed || any_cav)) continue;     // streamed
            if (M == 0) {
                double best_cost = 0.0;
                for (int m : ms_all) {  // cheapest tiles win; on a tie the larger lanes
                    if (m == 16 && (any_cav || any_chunky)) continue;
                    int c_k[kWave + 1] = {};
                    bool ok = true;
                    for (int64_t q = coff[r]; q < coff[r + 1] && ok; q++) {
                        if (is_small(csurf[q])) continue;
                        const int kk = (placed[csurf[q]].n + m - 1) / m;
                        ok = kk >= ((m == 8 && !any_cav && !mixed) ? 1 : 2) && kk <= kWave &&  // (single-lane surfaces: 8 nodes per lane, no cavities)
                             (cat[csurf[q]].m_ok & m_bit(m)) != 0;                              // (no-mass chunks whole in their lanes)
                        if (ok) c_k[kk]++;
                    }
                    if (!ok) continue;
                    const int nt_m = tiles_needed_packed(c_k) + small_tiles(n_small);
                    if (nt_m > max_tiles(m)) continue;
                    const double t = cluster_ns(m, nt_m);
                    if (M == 0 || t <= best_cost) { M = m; best_cost = t; }
                }
                if (M == 0) M = 4;
            }
            int nm = 0, cnt[kWave + 1] = {}, ne = 0;
            bool fits = true;
            for (int64_t q = coff[r]; q < coff[r + 1]; q++) {
                const Placed &pl = placed[csurf[q]];
                ne += (zone_of_side(csurf[q], 0) >= 0) + (zone_of_side(csurf[q], 1) >= 0);
                if (is_small(csurf[q])) continue;
                const int k = (pl.n + M - 1) / M;
                if (k > kWave || k < ((M == 8 && !any_cav && !mixed) ? 1 : 2) || !(cat[csurf[q]].m_ok & m_bit(M))) { fits = false; break; }
                cnt[k]++;
                nm |= cat[csurf[q]].nm;
            }
            const int nz = (int)czones[r].size();
            if (!fits || tiles_needed_packed(cnt) + small_tiles(n_small) > max_tiles(M) || nz > kFusedMaxZones ||
                ne > kFusedMaxEntries) {
                // Too large for one workgroup: a TEAM of up to kTeamMax workgroups of four wavefronts (layout.hpp). The
                // cluster's walls are dealt to the members room by room (by the smaller zone they face), so that most
                // zones are faced from one member only; a zone faced from several is balanced from their partial sums.
                static const bool teams_off = getenv("HEAT_AMD_NO_TEAMS") != nullptr;
                if (teams_off || opt.n_ranks > 1 || mixed || any_cav || any_chunky || nz > kTeamZones) continue;  // streamed
                int Mt = opt.nodes_per_lane;
                if (Mt == 0) {
                    double best = 0.0;
                    for (int m : ms_all) {
                        bool ok = true;
                        double c = 0.0;
                        for (int64_t q = coff[r]; q < coff[r + 1] && ok; q++) {
                            const int kk = (placed[csurf[q]].n + m - 1) / m;
                            ok = kk >= (m == 8 ? 1 : 2) && kk <= kWave && (cat[csurf[q]].m_ok & m_bit(m)) != 0;
                            c += padded_cost(placed[csurf[q]].n, m);
                        }
                        if (ok && (Mt == 0 || c < best)) { Mt = m; best = c; }
                    }
                    if (Mt == 0) continue;  // streamed
                }
                std::vector<int64_t> walls(csurf.begin() + coff[r], csurf.begin() + coff[r + 1]);
                auto room_of = [&](int64_t s) {
                    const int32_t zf = zone_of_side(s, 0), zb = zone_of_side(s, 1);
                    return zf < 0 ? zb : (zb < 0 ? zf : std::min(zf, zb));
                };
                std::stable_sort(walls.begin(), walls.end(), [&](int64_t x, int64_t y) { return room_of(x) < room_of(y); });
                struct Member { std::vector<int64_t> walls; std::vector<int32_t> zones; int cnt[kWave + 1] = {}; int ne = 0; int lanes = 0; };
                std::vector<Member> mem(1);
                bool ok = true;
                int nm_t = 0;
                // even shares: as few members as the lanes need, each filled to the same level (a member of one tile
                // beside two of four holds a workgroup slot for a quarter of the work)
                int64_t all_lanes = 0;
                for (int64_t s : walls) all_lanes += (placed[s].n + Mt - 1) / Mt;
                const int want_members = (int)std::max<int64_t>(2, (all_lanes + 4 * kWave - 1) / (4 * kWave));
                const int lane_target = (int)std::min<int64_t>(4 * kWave, (all_lanes + want_members - 1) / want_members + kWave / 4);
                for (int64_t s : walls) {
                    const int kk = (placed[s].n + Mt - 1) / Mt;
                    if (kk > kWave || kk < (Mt == 8 ? 1 : 2) || !(cat[s].m_ok & m_bit(Mt))) { ok = false; break; }
                    nm_t |= cat[s].nm;
                    for (int attempt = 0; attempt < 2; attempt++) {
                        Member &me = mem.back();
                        int zadd = 0;
                        for (int side = 0; side < 2; side++) {
                            const int32_t z = zone_of_side(s, side);
                            if (z >= 0 && std::find(me.zones.begin(), me.zones.end(), z) == me.zones.end() &&
                                !(side == 1 && z == zone_of_side(s, 0))) zadd++;
                        }
                        const int eadd = (zone_of_side(s, 0) >= 0) + (zone_of_side(s, 1) >= 0);
                        me.cnt[kk]++;
                        const bool fits_m = tiles_needed_packed(me.cnt) <= 4 && (int)me.zones.size() + zadd <= kFusedMaxZones &&
                                            me.ne + eadd <= kFusedMaxEntries &&
                                            (me.lanes + kk <= lane_target || (int)mem.size() >= want_members);
                        if (!fits_m) {
                            me.cnt[kk]--;
                            if (attempt == 1 || me.walls.empty()) { ok = false; break; }
                            mem.emplace_back();
                            continue;
                        }
                        me.walls.push_back(s);
                        me.ne += eadd;
                        me.lanes += kk;
                        for (int side = 0; side < 2; side++) {
                            const int32_t z = zone_of_side(s, side);
                            if (z >= 0 && std::find(me.zones.begin(), me.zones.end(), z) == me.zones.end()) me.zones.push_back(z);
                        }
                        break;
                    }
                    if (!ok || (int)mem.size() > kTeamMax) { ok = false; break; }
                }
                if (!ok || mem.size() < 2) continue;  // streamed
                if (!fuse_always && S > 8192) {
                    // tiles, plus what the exchange costs a member per sub-timestep (it waits ~3 us of a 512-workgroup chip)
                    constexpr double kTeamExchangeNs = 6.0;
                    double t = 0.0, bytes = 0.0;
                    for (const Member &me : mem) t += cluster_ns(Mt, tiles_needed_packed(me.cnt)) + kTeamExchangeNs;
                    for (int64_t s : walls) bytes += 32.0 * placed[s].n + 152.0;
                    if (t > 0.85 * (bytes / kStreamBytesPerNs + kZoneNs * nz)) continue;  // streamed
                }
                const int cls_t = fast_class(Mt, Category{0, nm_t, 0, 1, 7});
                // exchange slot of a zone = its number in the cluster; members that face it
                std::vector<uint32_t> zmask(nz, 0);
                auto slot_of = [&](int32_t z) { return (int)(std::lower_bound(czones[r].begin(), czones[r].end(), z) - czones[r].begin()); };
                for (size_t m = 0; m < mem.size(); m++)
                    for (int32_t z : mem[m].zones) zmask[slot_of(z)] |= 1u << m;
                for (size_t m = 0; m < mem.size(); m++) {
                    BlockPlan bp{cls_t, false, mem[m].zones, n_supers, (int)m, {}};
                    for (int32_t z : mem[m].zones) bp.zinfo.push_back((uint32_t)slot_of(z) | (zmask[slot_of(z)] << 16));
                    blocks.push_back(bp);
                    for (int64_t s : mem[m].walls) {
                        placed[s].blk = (int)blocks.size() - 1;
                        placed[s].cls = cls_t;
                        placed[s].k = (placed[s].n + Mt - 1) / Mt;
                    }
                }
                n_supers++;
                continue;
            }
            if (!fuse_always) {
                // No glazing in a fused workgroup: a window's no-mass loop re-evaluates its gas cavity every pass
                // (surface.rs:814) — a long serial chain the whole workgroup would wait for at every sub-timestep's
                // barrier (rooms with double glazing: 8x slower fused than streamed). Otherwise: tiles against bytes.
                int n_cav_small = 0;
                double bytes = 0.0;
                for (int64_t q = coff[r]; q < coff[r + 1]; q++) {
                    n_cav_small += cat[csurf[q]].kind == kSmallCav;
                    bytes += 32.0 * placed[csurf[q]].n + 152.0;
                }
                const int tiles = tiles_needed_packed(cnt) + small_tiles(n_small);
                // A small batch is bound by launches and latency, not by throughput: there the resident march wins
                // with any blocking factor (one launch per march call instead of two or more per sub-timestep).
                const bool small_batch = S <= 8192;
                if (n_cav_small > 0 || tiles == 0) continue;                                        // streamed
                if (!small_batch && cluster_ns(M, tiles) > 0.85 * (bytes / kStreamBytesPerNs + kZoneNs * nz))
                    continue;                                                                       // streamed
            }
            Category cc{0, mixed ? 1 : nm, (mixed ? (M < 16) : any_cav) ? 1 : 0, 1, 7};
            const int cls = fast_class(M, cc);
            Open &o = open[2 * cls + (mixed ? 1 : 0)];
            int merged[kWave + 1];
            for (int k = 0; k <= kWave; k++) merged[k] = o.cnt[k] + cnt[k];
            // Workgroups of four tiles are the target (two of them share a compute unit, so one's zone balance —
            // a short serial section — overlaps the other's stencil work): clusters are merged only up to four
            // tiles; a cluster that needs five to eight gets a workgroup of its own.
            if (o.blk < 0 || tiles_needed_packed(merged) + small_tiles(o.nsmall + n_small) > 4 || o.nz + nz > kFusedMaxZones ||
                o.ne + ne > kFusedMaxEntries) {
                o = Open();
                o.blk = new_block(cls, mixed);
                for (int k = 0; k <= kWave; k++) merged[k] = cnt[k];
            }
            for (int k = 0; k <= kWave; k++) o.cnt[k] = merged[k];
            o.nsmall += n_small;
            o.nz += nz;
            o.ne += ne;
            for (int32_t z : czones[r]) blocks[o.blk].zones.push_back(z);
            for (int64_t q = coff[r]; q < coff[r + 1]; q++) {
                Placed &pl = placed[csurf[q]];
                pl.blk = o.blk;
                if (is_small(csurf[q])) {
                    pl.cls = kSmallCav;  // (one kind of small tile inside workgroups: the cavity variant covers both)
                } else {
                    pl.cls = cls;
                    pl.k = (pl.n + M - 1) / M;
                }
            }
        }
        // surfaces that face no zone: any grouping will do; workgroups of up to 4 tiles of equal k
        std::vector<int64_t> lone;
        std::vector<uint8_t> lone_ok(S, 1);
        for (int64_t s = 0; s < S; s++)
            if (sroot[s] < 0 && fusable(s) && !is_small(s)) lone.push_back(s);
        for (int64_t s : lone) {
            Placed &pl = placed[s];
            int M = opt.nodes_per_lane;
            const bool cav = cat[s].ncav > 0;
            if (M == 0) {
                M = 4;
                for (int m : {8, 16})
                    if (!(m == 16 && (cav || cat[s].chunky)) && (cat[s].m_ok & m_bit(m)) && (pl.n + m - 1) / m >= ((m == 8 && !cav) ? 1 : 2) &&
                        fused_cost(pl.n, m) <= fused_cost(pl.n, M)) M = m;
            }
            int k = (pl.n + M - 1) / M;
            const bool gains = tile_ns(M) * k / kWave < 0.85 * (32.0 * pl.n + 152.0) / kStreamBytesPerNs;
            if (k > kWave || k < ((M == 8 && !cav) ? 1 : 2) || (M == 16 && (cav || cat[s].chunky)) || (cav && cat[s].chunky) || !(cat[s].m_ok & m_bit(M)) ||
                (!fuse_always && S > 8192 && !gains)) { lone_ok[s] = 0; continue; }
            pl.cls = fast_class(M, Category{0, cat[s].nm, cav ? 1 : 0, 1, 7});
            pl.k = k;
        }
        std::stable_sort(lone.begin(), lone.end(), [&](int64_t x, int64_t y) {
            if (placed[x].cls != placed[y].cls) return placed[x].cls < placed[y].cls;
            return placed[x].k < placed[y].k;
        });
        {
            int cur_cls = -1, cur_k = -1, cur_blk = -1, in_blk = 0;
            for (int64_t s : lone) {
                Placed &pl = placed[s];
                if (!lone_ok[s] || pl.cls >= kNumFast) continue;
                const int cap = 4 * (kWave / pl.k);
                if (pl.cls != cur_cls || pl.k != cur_k || in_blk >= cap) {
                    cur_cls = pl.cls; cur_k = pl.k; in_blk = 0;
                    cur_blk = new_block(pl.cls, false);
                }
                pl.blk = cur_blk;
                in_blk++;
            }
        }
    }
    if (fuse && !fuse_always && !blocks.empty()) {
        // A SMALL batch (bound by launches, not by throughput) is fused only as a whole: a streamed remainder would
        // still pay its launches every sub-timestep (2 000 clustered walls with 733 of them fused: 89 us per
        // sub-timestep against 64 all streamed). A large batch fuses the clusters that gain and streams the rest
        // AFTER them in the same march call, on the same stream (heat_batch_march_resident): the two kinds of kernels
        // never share the chip — side by side they did badly (1 M clustered walls with 205 000 fused: 260 against 222
        // all streamed) —, so every fused cluster keeps its gain; below a tenth of the batch's nodes the fused
        // launch's own ramp and tail are not worth it.
        double fused_nodes = 0.0, all_nodes = 0.0;
        bool remainder = false;
        for (int64_t s = 0; s < S; s++) {
            all_nodes += placed[s].n;
            if (placed[s].blk >= 0) fused_nodes += placed[s].n;
            else remainder = true;
        }
        static const double min_share = getenv("HEAT_AMD_FUSE_MIN_SHARE") ? atof(getenv("HEAT_AMD_FUSE_MIN_SHARE")) : 0.1;
        if (S <= 8192 ? remainder : fused_nodes < min_share * all_nodes) {
            placed = placed_streamed;
            blocks.clear();
        } else if (remainder) {
            // the streamed remainder keeps the blocking factors of the all-streamed plan
            for (int64_t s = 0; s < S; s++)
                if (placed[s].blk < 0) placed[s] = placed_streamed[s];
        }
    }
    for (int64_t s = 0; s < S; s++) {
        const int cls = placed[s].cls;
        p.class_counts[cls < kNumFast ? cls / 6 : (cls < kGeneral ? 3 : 4)]++;
        if (cls < kNumFast && kFastPAL[cls]) p.n_palette++;
        if (placed[s].blk >= 0) p.n_fused_surfaces++;
    }

    // ---- order: class, then streamed surfaces before the fused workgroups, then lanes per surface ----
    // Small surfaces with a gas cavity are grouped by the branch of the Nusselt correlation their tilt selects
    // (gas.rs:197-315: five ranges of the cavity angle): the lanes of a wavefront then take the same branch instead of
    // the wavefront running all of them one after the other.
    std::vector<uint8_t> tilt_key(S, 0);
    if (d->seg_cavity && d->n_cavities > 0)
        for (int64_t s = 0; s < S; s++) {
            if (placed[s].cls != kSmallCav && !(placed[s].cls < kNumFast && kFastCAV[placed[s].cls])) continue;
            const int64_t o = d->node_offset[s];
            for (int i = 0; i < placed[s].n; i++)
                if (d->seg_cavity[o + i] >= 0) {
                    double g = std::fmod(d->cavities[d->seg_cavity[o + i]].angle, 3.14159265358979323846);
                    if (g > 1.5707963267948966) g = 3.14159265358979323846 - g;  // the kernel flips the angle by the sign of dT
                    const double deg = g * (180.0 / 3.14159265358979323846);
                    tilt_key[s] = deg < 59.5 ? 0 : (deg < 60.5 ? 1 : (deg < 89.5 ? 2 : 3));
                    break;
                }
        }
    std::vector<int64_t> order(S);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) {
        const Placed &a = placed[x], &c = placed[y];
        if (a.cls != c.cls) return a.cls < c.cls;
        if (a.blk != c.blk) return a.blk < c.blk;
        if (a.cls < kNumFast && a.k != c.k) return a.k < c.k;
        if (tilt_key[x] != tilt_key[y]) return tilt_key[x] < tilt_key[y];
        if (a.cls < kNumFast) return false;
        return a.n < c.n;
    });

    // ---- tiles ----
    std::vector<FastTile> fast_tiles[kNumFast];
    std::vector<GeneralTile> gen_tiles;
    std::vector<int64_t> dev_of(S);            // original -> device surface
    std::vector<int64_t> node0_index(S), nodeN_index(S);  // index of first / last node in the T buffer
    std::vector<int64_t> orig_of(S);
    int64_t node_cursor = 0, scratch_cursor = 0;
    int64_t dcur = 0;
    size_t pos = 0;
    struct NodeMap { int64_t base; int Lk; int k; int M; int g; int tile; int lane0; };  // per device surface
    std::vector<int> blk_first_tile(blocks.size(), -1), blk_n_tiles(blocks.size(), 0);
    std::vector<int> blk_first_small(blocks.size(), -1), blk_n_small(blocks.size(), 0);
    std::vector<NodeMap> nmap(S);
    int prev_cls = -1;
    while (pos < (size_t)S) {
        const Placed &p0 = placed[order[pos]];
        if (p0.cls != prev_cls) node_cursor = (node_cursor + 15) / 16 * 16;  // class bytes are loaded 4/8/16 at a time
        prev_cls = p0.cls;
        if (p0.cls < kNumFast) {
            const int M = kFastM[p0.cls], k = p0.k;
            // A workgroup of the cluster-resident march whose surfaces differ in their lane counts packs them into
            // its wavefronts one after the other (mixed tiles, lane table behind the tile's class bytes); everything
            // else keeps tiles of one lane count, floor(64 / k) surfaces each.
            bool mixed_block = false;
            if (p0.blk >= 0) {
                for (size_t q = pos; q < (size_t)S && placed[order[q]].cls == p0.cls && placed[order[q]].blk == p0.blk; q++)
                    mixed_block = mixed_block || placed[order[q]].k != k;
            }
            const int Gmax = kWave / k;
            size_t end = pos;
            int lanes = 0;
            if (mixed_block) {
                while (end < (size_t)S && placed[order[end]].cls == p0.cls && placed[order[end]].blk == p0.blk &&
                       lanes + placed[order[end]].k <= kWave) {
                    lanes += placed[order[end]].k;
                    end++;
                }
            } else {
                while (end < (size_t)S && placed[order[end]].cls == p0.cls && placed[order[end]].k == k &&
                       placed[order[end]].blk == p0.blk && (int)(end - pos) < Gmax)
                    end++;
            }
            const int Lk = mixed_block ? lanes : Gmax * k;
            bool all_full = true;
            for (size_t q = pos; q < end; q++) all_full = all_full && (placed[order[q]].n == placed[order[q]].k * M);
            FastTile t;
            t.node_base = node_cursor;
            t.surf_base = (int32_t)dcur;
            bool any_chunky = false;
            for (size_t q = pos; q < end; q++) any_chunky = any_chunky || cat[order[q]].chunky;
            t.k = (int16_t)((mixed_block ? (Lk | kTileMixedBit) : k) | (all_full ? 0x100 : 0) | (any_chunky ? kTileChunkyBit : 0));
            t.G = (int16_t)(end - pos);
            const int tile_index = (int)fast_tiles[p0.cls].size();
            fast_tiles[p0.cls].push_back(t);
            if (p0.blk >= 0) {
                if (blk_first_tile[p0.blk] < 0) blk_first_tile[p0.blk] = tile_index;
                blk_n_tiles[p0.blk]++;
            } else {
                p.n_stream_tiles[p0.cls] = tile_index + 1;  // streamed tiles come first in every class
            }
            int lane0 = 0;
            for (size_t q = pos; q < end; q++) {
                const int64_t s = order[q];
                const int ks = placed[s].k;
                dev_of[s] = dcur;
                orig_of[dcur] = s;
                nmap[dcur] = NodeMap{node_cursor, Lk, ks, M, (int)(q - pos), tile_index, lane0};
                lane0 += ks;
                dcur++;
            }
            node_cursor += (int64_t)M * Lk + (mixed_block ? kLaneTableSlots : 0);
            pos = end;
        } else {
            if (gen_tiles.empty()) p.gen_base = node_cursor;
            size_t end = pos;
            while (end < (size_t)S && end < pos + (size_t)kWave && placed[order[end]].cls == p0.cls &&
                   placed[order[end]].blk == p0.blk)
                end++;
            if (p0.cls < kGeneral) p.n_small_tiles++;
            if (p0.cls == kSmall) p.n_small_plain_tiles++;
            if (p0.cls == kSmallCav && p0.blk < 0) p.n_smallcav_stream_tiles++;
            if (p0.blk >= 0) {
                if (blk_first_small[p0.blk] < 0) blk_first_small[p0.blk] = (int)gen_tiles.size();
                blk_n_small[p0.blk]++;
            }
            int n_max = 0;
            for (size_t q = pos; q < end; q++) n_max = std::max(n_max, placed[order[q]].n);
            GeneralTile t;
            t.node_base = node_cursor;
            t.surf_base = (int32_t)dcur;
            t.G = (int32_t)(end - pos);
            t.n_max = n_max;
            t.pad = 0;
            t.scratch_base = scratch_cursor;
            gen_tiles.push_back(t);
            for (size_t q = pos; q < end; q++) {
                const int64_t s = order[q];
                dev_of[s] = dcur;
                orig_of[dcur] = s;
                nmap[dcur] = NodeMap{node_cursor, kWave, 1, 0, (int)(q - pos), (int)gen_tiles.size() - 1, (int)(q - pos)};
                dcur++;
            }
            node_cursor += (int64_t)n_max * kWave;
            scratch_cursor += (int64_t)kScratchArrays * n_max * kWave;
            pos = end;
        }
    }
    if (gen_tiles.empty()) p.gen_base = node_cursor;
    p.node_slots = node_cursor;
    if (node_cursor >= (int64_t)1 << 32) return failp(err, HEAT_E_SIZE, "batch too large: %lld node slots", (long long)node_cursor);

    auto node_index = [&](int64_t dsurf, int i) -> int64_t {
        const NodeMap &m = nmap[dsurf];
        if (m.M == 0) return m.base + (int64_t)i * kWave + m.g;
        const int lane = m.lane0 + i / m.M, j = i % m.M;
        return m.base + ((int64_t)(j >> 1) * m.Lk + lane) * 2 + (j & 1);
    };

    // ---- per-node constants ----
    std::vector<double> hV(node_cursor, 0.0), hU(node_cursor, 0.0);
    std::vector<uint8_t> hCls(p.n_palette ? node_cursor : 0, 0);
    // the palette width of the batch (layout.hpp): narrow unless a palette-form wall needs more entries
    static const bool tiny_off = getenv("HEAT_AMD_NO_TINY_PAL") != nullptr;  // measurement
    p.pal_stride = tiny_off ? kPalNarrow : kPalTiny;
    for (int64_t s = 0; s < S; s++)
        if (placed[s].cls < kNumFast && kFastPAL[placed[s].cls]) {
            if (cat[s].wide) p.pal_stride = kPal;
            else if (!cat[s].tiny) p.pal_stride = std::max(p.pal_stride, kPalNarrow);
        }
    p.pal_ubase = (p.pal_stride == kPal) ? kPalV : (p.pal_stride == kPalNarrow ? kPalVNarrow : kPalVTiny);
    const int pstride = p.pal_stride, ubase = p.pal_ubase;
    std::vector<double> hPal(p.n_palette ? (size_t)S * pstride : 0, 0.0);
    const int64_t gen_slots = node_cursor - p.gen_base;
    std::vector<double> hAf(gen_slots, 0.0), hAb(gen_slots, 0.0), hMass(gen_slots, 0.0);
    std::vector<int32_t> hCav(gen_slots, -1);
    for (int64_t dd = 0; dd < S; dd++) {
        const int64_t s = orig_of[dd];
        const int64_t o = d->node_offset[s];
        const int n = placed[s].n;
        const bool gen = placed[s].cls >= kNumFast;
        const bool pal = !gen && kFastPAL[placed[s].cls];
        int nv = 1, nu = 1;
        double *pp = pal ? &hPal[(size_t)dd * pstride] : nullptr;
        for (int i = 0; i < n; i++) {
            const int64_t idx = node_index(dd, i);
            const double mass = d->mass[o + i];
            hV[idx] = (mass >= kMassThreshold) ? d->dt / mass : 0.0;  // dt / C, surface.rs:172
            const bool cav = d->seg_cavity && d->n_cavities > 0 && d->seg_cavity[o + i] >= 0;
            hU[idx] = cav ? 0.0 : d->uvalue[o + i];
            if (pal) {
                int vc = -1, uc = -1;
                for (int q = 0; q < nv; q++) if (pp[q] == hV[idx]) vc = q;
                if (vc < 0) { vc = nv; pp[nv++] = hV[idx]; }
                for (int q = 0; q < nu; q++) if (pp[ubase + q] == hU[idx]) uc = q;
                if (uc < 0) { uc = nu; pp[ubase + nu++] = hU[idx]; }
                const NodeMap &m = nmap[dd];
                const int lane = m.lane0 + i / m.M, j = i % m.M;
                hCls[m.base + (int64_t)lane * m.M + j] = (uint8_t)(vc | (uc << kPalUShift));
            }
            if (gen) {
                const int64_t gi = idx - p.gen_base;
                hAf[gi] = d->front_alpha[o + i];
                hAb[gi] = d->back_alpha[o + i];
                hMass[gi] = mass;
                hCav[gi] = cav ? d->seg_cavity[o + i] : -1;
            }
        }
        node0_index[s] = node_index(dd, 0);
        nodeN_index[s] = node_index(dd, n - 1);
    }

    // ---- no-mass chunk marks: V index 14 / 15 in the class byte of a chunk's first node = one / two nodes (kernels.hip) ----
    if (!hCls.empty())
        for (int64_t dd = 0; dd < S; dd++) {
            const int64_t s = orig_of[dd];
            if (!(placed[s].cls < kNumFast && kFastPAL[placed[s].cls] && cat[s].nm)) continue;
            const int64_t o = d->node_offset[s];
            const NodeMap &m = nmap[dd];
            for (int i = 0; i < placed[s].n;) {
                if (d->mass[o + i] >= kMassThreshold) { i++; continue; }
                int e = i;
                while (e < placed[s].n && d->mass[o + e] < kMassThreshold) e++;
                hCls[m.base + (int64_t)(m.lane0 + i / m.M) * m.M + i % m.M] |= (uint8_t)(kPalVMark1 - 1 + (e - i));  // (its V index is 0)
                i = e;
            }
        }

    // ---- lane tables of the mixed tiles: behind the tile's class bytes, one 16-bit entry per lane ----
    for (int c = 0; c < kNumFast; c++)
        for (const FastTile &ft : fast_tiles[c]) {
            if (!(ft.k & kTileMixedBit)) continue;
            const int M = kFastM[c], Lk = ft.k & 0xff;
            uint8_t *tab = &hCls[ft.node_base + (int64_t)M * Lk];
            int lane = 0;
            for (int g = 0; g < ft.G; g++) {
                const int ks = nmap[ft.surf_base + g].k;
                for (int seg = 0; seg < ks; seg++, lane++) {
                    const uint16_t e = (uint16_t)(g | (seg << 6) | (seg == ks - 1 ? kLaneLastBit : 0));
                    tab[2 * lane] = (uint8_t)(e & 0xff);
                    tab[2 * lane + 1] = (uint8_t)(e >> 8);
                }
            }
        }

    // ---- cavity references of the CAV fast classes ----
    std::vector<int32_t> hCavRef;
    {
        bool any = false;
        for (int64_t s = 0; s < S; s++) any = any || (placed[s].cls < kNumFast && kFastCAV[placed[s].cls]);
        if (any) {
            hCavRef.assign((size_t)4 * S, -1);
            for (int64_t dd = 0; dd < S; dd++) {
                const int64_t s = orig_of[dd];
                if (!(placed[s].cls < kNumFast && kFastCAV[placed[s].cls])) continue;
                const int64_t o = d->node_offset[s];
                int r = 0;
                for (int i = 0; i < placed[s].n && r < 2 && d->seg_cavity && d->n_cavities > 0; i++)
                    if (d->seg_cavity[o + i] >= 0) {
                        hCavRef[4 * dd + 2 * r] = i;
                        hCavRef[4 * dd + 2 * r + 1] = d->seg_cavity[o + i];
                        r++;
                    }
            }
        }
    }

    // ---- per-side records (device order) ----
    std::vector<int32_t> hMeta(S);
    std::vector<SideConst> hSide(2 * S);
    std::vector<double> hAlpha(2 * S, 1.0);
    std::vector<double> hFix;
    std::vector<int64_t> hFirst(S), hSlots(8 * S);
    const bool has_fix = d->front_hs_fix != nullptr;
    if (has_fix) hFix.resize(2 * S);
    for (int64_t dd = 0; dd < S; dd++) {
        const int64_t s = orig_of[dd];
        const int64_t o = d->node_offset[s];
        const int n = placed[s].n;
        hMeta[dd] = n;
        const double cos_tilt = d->cos_tilt[s];
        // is_windward: only tilted surfaces test the wind direction (surface.rs:38)
        const int always_windward = (std::fabs(cos_tilt) < 0.98) ? 0 : 4;
        // forced convection: 2.537 * Wf * Rf * sqrt(P * V_z / A), Rf = COEFFICIENTS[1] = 1.67, V_z = wind * modifier
        // (convection.rs:151-168; surface.rs:646,691). Wf and sqrt(wind) are applied per sub-timestep.
        const double forced = 2.537 * 1.67 * std::sqrt(d->perimeter[s] * d->wind_modifier[s] / d->area[s]);
        for (int side = 0; side < 2; side++) {
            SideConst c;
            const int kind = side ? d->back_kind[s] : d->front_kind[s];
            const int peer = side ? d->front_kind[s] : d->back_kind[s];  // the surface's other side (layout.hpp)
            c.kind_n = kind | always_windward | ((peer & 3) << 4) | (n << 16);
            c.zone = kind == HEAT_BOUNDARY_SPACE ? (side ? d->back_zone[s] : d->front_zone[s]) : 0;
            c.ambient = side ? d->back_ambient[s] : d->front_ambient[s];
            c.emis = side ? d->back_emissivity[s] : d->front_emissivity[s];
            c.alpha = side ? d->back_alpha[o + n - 1] : d->front_alpha[o];
            // front Outdoor flips the sign (surface.rs:652); back Outdoor does not (surface.rs:689-696)
            c.cos_eff = (side == 0 && kind == HEAT_BOUNDARY_OUTDOOR) ? -cos_tilt : cos_tilt;
            hAlpha[(int64_t)side * S + dd] = 1.0;
            if (placed[s].cls < kNumFast) {
                // fast classes: the absorptance goes into SideDyn::solar at upload; the two slots carry the TARP
                // natural-convection coefficients of convection.rs:87-110 with the tilt-dependent division done here
                hAlpha[(int64_t)side * S + dd] = c.alpha;
                const double ce = c.cos_eff, act = std::fabs(ce);
                const double up = 9.482 / (7.238 - act), down = 1.81 / (1.382 + act);
                double pos, neg;  // air warmer / colder than the surface
                if (ce != ce) { pos = neg = ce; }
                else if (act < 1e-3) { pos = neg = 1.31; }
                else if (ce > 0.) { pos = up; neg = down; }
                else { pos = down; neg = up; }
                c.cos_eff = pos;
                c.alpha = neg;
            }
            // (a Space-facing side carries the surface's area here: what its contribution to the zone is weighted with)
            c.forced = kind == HEAT_BOUNDARY_OUTDOOR ? forced : (kind == HEAT_BOUNDARY_SPACE ? d->area[s] : 0.0);
            if (side == 1 && kind == HEAT_BOUNDARY_AMBIENT) {
                // A back side facing an ambient temperature takes t_front for its radiant temperature
                // (surface.rs:672-686): the FRONT side's boundary source travels in this record's unused slots, so that
                // the kernels need no second record (a dependent load in front of every tile that holds such a wall)
                if (peer == HEAT_BOUNDARY_SPACE) c.zone = d->front_zone[s];
                if (peer == HEAT_BOUNDARY_AMBIENT) c.forced = d->front_ambient[s];
            }
            c.nx = d->normal_x[s];
            c.ny = d->normal_y[s];
            hSide[(int64_t)side * S + dd] = c;
            if (has_fix) hFix[(int64_t)side * S + dd] = side ? d->back_hs_fix[s] : d->front_hs_fix[s];
        }
        hFirst[dd] = d->first_node_slot[s];
        const int64_t *src[8] = {d->hs_front_slot, d->hs_back_slot, d->flow_front_slot, d->flow_back_slot,
                                 d->solar_front_slot, d->solar_back_slot, d->ir_front_slot, d->ir_back_slot};
        for (int a = 0; a < 8; a++) hSlots[(int64_t)a * S + dd] = src[a][s];
    }

    // ---- zone contribution lists, in the reference's order (model.rs:562-585) ----
    std::vector<int64_t> zoff(Z + 1, 0);
    for (int64_t s = 0; s < S; s++) {
        if (d->front_kind[s] == HEAT_BOUNDARY_SPACE) zoff[d->front_zone[s] + 1]++;
        if (d->back_kind[s] == HEAT_BOUNDARY_SPACE) zoff[d->back_zone[s] + 1]++;
    }
    for (int64_t z = 0; z < Z; z++) zoff[z + 1] += zoff[z];
    std::vector<ZoneEntry> zent(Z > 0 ? zoff[Z] : 0);
    {
        std::vector<int64_t> cur(zoff.begin(), zoff.end() - (Z >= 0 ? 1 : 0));
        // a Space-facing side keeps its position in the zone's list where an ambient side keeps its temperature
        auto put_pos = [&](int64_t rec, int64_t pos) {
            static_assert(sizeof(int64_t) == sizeof(double), "entry position is stored in the ambient slot");
            memcpy(&hSide[rec].ambient, &pos, sizeof pos);
        };
        for (int64_t s = 0; s < S; s++) {
            if (d->front_kind[s] == HEAT_BOUNDARY_SPACE) {
                ZoneEntry e{(uint32_t)node0_index[s], (uint32_t)dev_of[s], d->area[s]};
                put_pos(dev_of[s], cur[d->front_zone[s]]);
                zent[cur[d->front_zone[s]]++] = e;
            }
            if (d->back_kind[s] == HEAT_BOUNDARY_SPACE) {
                ZoneEntry e{(uint32_t)nodeN_index[s], (uint32_t)(S + dev_of[s]), d->area[s]};
                put_pos(S + dev_of[s], cur[d->back_zone[s]]);
                zent[cur[d->back_zone[s]]++] = e;
            }
        }
    }

    // ---- cluster-resident march: workgroup tables ----
    std::vector<int32_t> fz, fz_eoff(1, 0);
    std::vector<uint16_t> fent;
    std::vector<double> side_area(2 * S, 0.0);
    std::vector<int16_t> side_lz(2 * S, -1);
    p.zone_block.assign(Z, -1);
    p.any_fused = false;
    std::vector<uint32_t> fteam;
    if (!blocks.empty()) {
        std::vector<int> blk_fw(blocks.size(), 4);
        std::vector<int32_t> blk_first_zone(blocks.size(), 0);
        for (size_t bi = 0; bi < blocks.size(); bi++) {
            if (blk_n_tiles[bi] + blk_n_small[bi] <= 0) continue;  // (an empty plan: cannot happen, kept harmless)
            blk_fw[bi] = blk_n_tiles[bi] + blk_n_small[bi] <= 4 ? 4 : 8;
            blk_first_zone[bi] = (int32_t)fz.size();
            for (size_t j = 0; j < blocks[bi].zones.size(); j++) {
                const int32_t z = blocks[bi].zones[j];
                p.zone_block[z] = (int32_t)bi;
                fz.push_back(z);
                fteam.push_back(blocks[bi].super >= 0 ? blocks[bi].zinfo[j] : 0u);
            }
        }
        // (a zone belongs to ONE workgroup's list, except in a team, where every member that faces it lists it: the
        // place of zone z in workgroup bi's list is looked up there — the lists are short)
        auto local_zone = [&](int bi, int32_t z) -> int32_t {
            const std::vector<int32_t> &zl = blocks[bi].zones;
            return (int32_t)(std::find(zl.begin(), zl.end(), z) - zl.begin());
        };
        // contributions in the reference's order inside each zone (model.rs:562-585): counting sort by fused zone
        std::vector<int32_t> cnt(fz.size() + 1, 0);
        auto side_zone = [&](int64_t s, int side) -> int32_t {
            const int kind = side ? d->back_kind[s] : d->front_kind[s];
            const int32_t z = kind == HEAT_BOUNDARY_SPACE ? (side ? d->back_zone[s] : d->front_zone[s]) : -1;
            return (z >= 0 && placed[s].blk >= 0) ? z : -1;
        };
        for (int64_t s = 0; s < S; s++)
            for (int side = 0; side < 2; side++) {
                const int32_t z = side_zone(s, side);
                if (z >= 0) cnt[blk_first_zone[placed[s].blk] + local_zone(placed[s].blk, z) + 1]++;
            }
        for (size_t i = 0; i < fz.size(); i++) cnt[i + 1] += cnt[i];
        fz_eoff.assign(cnt.begin(), cnt.end());
        fent.resize(cnt[fz.size()]);
        std::vector<int32_t> cur(cnt.begin(), cnt.end() - 1);
        for (int64_t s = 0; s < S; s++)
            for (int side = 0; side < 2; side++) {
                const int32_t z = side_zone(s, side);
                if (z < 0) continue;
                const int bi = placed[s].blk;
                const NodeMap &m = nmap[dev_of[s]];
                // fast-path surfaces: the first / last lane of the surface owns the side; small surfaces: their lane
                const int lane = (m.M == 0) ? m.g : (side ? (m.lane0 + m.k - 1) : m.lane0);
                const int wave_in_block = (m.M == 0) ? blk_n_tiles[bi] + (m.tile - blk_first_small[bi])
                                                     : (m.tile - blk_first_tile[bi]);
                const uint32_t slot = (uint32_t)(side * kWave * blk_fw[bi] + wave_in_block * kWave + lane);
                const int32_t lz = local_zone(bi, z);
                fent[cur[blk_first_zone[bi] + lz]++] = (uint16_t)slot;
                side_lz[(int64_t)side * S + dev_of[s]] = (int16_t)lz;
            }
        int cur_super = -1;
        for (size_t bi = 0; bi < blocks.size(); bi++) {
            if (blk_n_tiles[bi] + blk_n_small[bi] <= 0) continue;
            FusedBlock fb{std::max(blk_first_tile[bi], 0), blk_n_tiles[bi], blk_first_zone[bi],
                          (int32_t)blocks[bi].zones.size(), std::max(blk_first_small[bi], 0), blk_n_small[bi]};
            if (blocks[bi].super >= 0) {
                // a team's members follow each other (they were planned in a row): one FusedSuper per team
                std::vector<FusedBlock> &tb = p.team_blocks[blocks[bi].cls];
                if (blocks[bi].super != cur_super) {
                    cur_super = blocks[bi].super;
                    p.team_supers[blocks[bi].cls].push_back(FusedSuper{(int32_t)tb.size(), 0});
                }
                p.team_supers[blocks[bi].cls].back().n_members++;
                tb.push_back(fb);
            } else {
                p.fblocks[blocks[bi].cls][(blk_fw[bi] == 4 ? 0 : 1) + (blocks[bi].mixed ? 2 : 0)].push_back(fb);
            }
            p.any_fused = true;
        }
    }
    p.team_zinfo = std::move(fteam);
    // zones this batch's surfaces touch (for sharded batches)
    p.touched.assign(Z, 0);
    for (int64_t z = 0; z < Z; z++) p.touched[z] = zoff[z + 1] > zoff[z] ? 1 : 0;

    // ---- what the plan hands over ----
    for (int c = 0; c < kNumFast; c++) p.fast_tiles[c] = std::move(fast_tiles[c]);
    p.gen_tiles = std::move(gen_tiles);
    p.scratch_slots = scratch_cursor;
    p.fzones = std::move(fz);
    p.fzone_eoff = std::move(fz_eoff);
    p.fslots = std::move(fent);
    for (int64_t dd = 0; dd < S; dd++) side_area[dd] = side_area[S + dd] = d->area[orig_of[dd]];
    p.side_area = std::move(side_area);
    p.side_lzone = std::move(side_lz);
    p.V = std::move(hV);
    p.U = std::move(hU);
    p.cls = std::move(hCls);
    p.pal = std::move(hPal);
    p.alpha_f = std::move(hAf);
    p.alpha_b = std::move(hAb);
    p.mass = std::move(hMass);
    p.cav_idx = std::move(hCav);
    p.cavref = std::move(hCavRef);
    p.cavs.resize(d->n_cavities);
    for (int64_t c = 0; c < d->n_cavities; c++) {
        const heat_cavity &x = d->cavities[c];
        if (x.gas < 0 || x.gas > 3) return failp(err, HEAT_E_INVALID_ARG, "cavity %lld: unknown gas", (long long)c);
        p.cavs[c] = CavityDev{x.thickness, x.height, x.angle, x.eout, x.ein, x.gas, 0};
    }
    p.meta = std::move(hMeta);
    p.side = std::move(hSide);
    p.side_alpha = std::move(hAlpha);
    p.hs_fix = std::move(hFix);
    p.first_slot = std::move(hFirst);
    p.slots = std::move(hSlots);
    p.dev_of = std::move(dev_of);
    p.orig_of = std::move(orig_of);
    p.zone_off = std::move(zoff);
    p.zone_entries = std::move(zent);
    p.zone_slot.assign(Z, 0);
    p.zone_vol.assign(Z, 0.0);
    for (int64_t z = 0; z < Z; z++) { p.zone_slot[z] = d->zone_slot[z]; p.zone_vol[z] = d->zone_volume[z]; }
    p.h_first_slot.assign(d->first_node_slot, d->first_node_slot + S);
    p.h_node_count.resize(S);
    for (int64_t s = 0; s < S; s++) p.h_node_count[s] = placed[s].n;
    p.h_out_slots[0].assign(d->hs_front_slot, d->hs_front_slot + S);
    p.h_out_slots[1].assign(d->hs_back_slot, d->hs_back_slot + S);
    p.h_out_slots[2].assign(d->flow_front_slot, d->flow_front_slot + S);
    p.h_out_slots[3].assign(d->flow_back_slot, d->flow_back_slot + S);
    {   // no-mass pass counters: one slot per tile of the NM fast classes, one per lane of the general-layout tiles
        size_t n = 0;
        for (int c = 0; c < kNumFast; c++) {
            p.nm_count_base[c] = n;
            if (kFastNM[c]) n += p.fast_tiles[c].size();
        }
        p.nm_count_base[kNumFast] = n;
        n += p.gen_tiles.size() * (size_t)kWave;
        p.n_nm_counters = std::max<size_t>(n, 1);
    }
    return HEAT_OK;
}

// ---------------------------------------------------------------------------
// Zone-connected clusters: zones joined by a Space/Space wall belong together (model.rs:556-590).
void find_clusters(const heat_batch_desc *d, std::vector<int32_t> &cluster_of_surface,
                   std::vector<int32_t> &cluster_of_zone, int32_t &n_clusters) {
    const int64_t S = d->n_surfaces, Z = d->n_zones;
    std::vector<int32_t> uf(Z);
    std::iota(uf.begin(), uf.end(), 0);
    auto find = [&](int32_t x) {
        while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; }
        return x;
    };
    auto zone_of_side = [&](int64_t s, int side) -> int32_t {
        const int kind = side ? d->back_kind[s] : d->front_kind[s];
        return kind == HEAT_BOUNDARY_SPACE ? (side ? d->back_zone[s] : d->front_zone[s]) : -1;
    };
    for (int64_t s = 0; s < S; s++) {
        const int32_t zf = zone_of_side(s, 0), zb = zone_of_side(s, 1);
        if (zf >= 0 && zb >= 0) {
            const int32_t a = find(zf), c = find(zb);
            if (a != c) uf[std::max(a, c)] = std::min(a, c);
        }
    }
    cluster_of_zone.assign(Z, -1);
    n_clusters = 0;
    for (int64_t z = 0; z < Z; z++) {  // roots are the smallest zone of their cluster: ids follow zone order
        const int32_t r = find((int32_t)z);
        if (cluster_of_zone[r] < 0) cluster_of_zone[r] = n_clusters++;
        cluster_of_zone[z] = cluster_of_zone[r];
    }
    cluster_of_surface.assign(S, -1);
    for (int64_t s = 0; s < S; s++) {
        const int32_t zf = zone_of_side(s, 0), zb = zone_of_side(s, 1);
        const int32_t z = zf >= 0 ? zf : zb;
        if (z >= 0) cluster_of_surface[s] = cluster_of_zone[z];
    }
}

// ---------------------------------------------------------------------------
// the cluster-resident march runs surfaces of one lane in the 8-node classes without cavities only (kernels.hip)
static bool single_lane_class(int c) { return kFastM[c] == 8 && !kFastCAV[c]; }

int check_plan(const Plan &p, const heat_batch_desc *d, std::string &err) {
#define PLAN_REQUIRE(cond, ...) \
    do { if (!(cond)) return failp(err, HEAT_E_SIZE, "plan check failed: " __VA_ARGS__); } while (0)
    const int64_t S = p.n_surf, Z = p.n_zones;
    PLAN_REQUIRE(S == d->n_surfaces && Z == d->n_zones, "counts");
    PLAN_REQUIRE((int64_t)p.dev_of.size() == S && (int64_t)p.orig_of.size() == S, "permutation size");
    for (int64_t s = 0; s < S; s++) {
        PLAN_REQUIRE(p.dev_of[s] >= 0 && p.dev_of[s] < S, "dev_of[%lld] = %lld", (long long)s, (long long)p.dev_of[s]);
        PLAN_REQUIRE(p.orig_of[p.dev_of[s]] == s, "orig_of is not the inverse of dev_of at %lld", (long long)s);
    }
    PLAN_REQUIRE((int64_t)p.side.size() == 2 * S && (int64_t)p.meta.size() == S, "side / meta size");
    PLAN_REQUIRE((int64_t)p.V.size() == p.node_slots && (int64_t)p.U.size() == p.node_slots, "V / U size");
    PLAN_REQUIRE(p.cls.empty() || (int64_t)p.cls.size() == p.node_slots, "class bytes size");
    PLAN_REQUIRE((p.pal_stride == kPalTiny && p.pal_ubase == kPalVTiny) || (p.pal_stride == kPalNarrow && p.pal_ubase == kPalVNarrow) ||
                     (p.pal_stride == kPal && p.pal_ubase == kPalV),
                 "palette stride %d, U entries from %d", p.pal_stride, p.pal_ubase);
    PLAN_REQUIRE(p.pal.empty() || (int64_t)p.pal.size() == S * p.pal_stride, "palette size");
    // tiles: every device surface in exactly one tile; node ranges inside the buffers and disjoint
    std::vector<uint8_t> covered(S, 0);
    std::vector<std::pair<int64_t, int64_t>> ranges;
    for (int c = 0; c < kNumFast; c++) {
        const int M = kFastM[c];
        PLAN_REQUIRE(p.n_stream_tiles[c] >= 0 && p.n_stream_tiles[c] <= (int)p.fast_tiles[c].size(), "class %d: streamed tiles", c);
        for (size_t t = 0; t < p.fast_tiles[c].size(); t++) {
            const FastTile &ft = p.fast_tiles[c][t];
            const bool mixed = (ft.k & kTileMixedBit) != 0;
            const int k = ft.k & 0xff;  // lanes per surface; mixed: lanes of the tile
            PLAN_REQUIRE(k >= 1 && k <= kWave && ft.G >= 1 && (mixed ? ft.G <= k : ft.G * k <= kWave), "class %d tile %zu: k %d G %d", c, t, k, (int)ft.G);
            const int Lk = mixed ? k : (kWave / k) * k;
            const int64_t extent = (int64_t)M * Lk + (mixed ? kLaneTableSlots : 0);
            PLAN_REQUIRE(ft.node_base >= 0 && ft.node_base + extent <= p.node_slots && ft.node_base % 2 == 0,
                         "class %d tile %zu: node range", c, t);
            PLAN_REQUIRE(ft.node_base + extent <= p.gen_base, "class %d tile %zu reaches into the general group", c, t);
            ranges.push_back({ft.node_base, ft.node_base + extent});
            PLAN_REQUIRE(ft.surf_base >= 0 && ft.surf_base + ft.G <= S, "class %d tile %zu: surfaces", c, t);
            PLAN_REQUIRE(!mixed || (kFastPAL[c] && !p.cls.empty()), "mixed tile outside the palette classes");
            int lane0 = 0;
            for (int g = 0; g < ft.G; g++) {
                const int64_t dd = ft.surf_base + g;
                PLAN_REQUIRE(!covered[dd], "device surface %lld in two tiles", (long long)dd);
                covered[dd] = 1;
                const int n = p.meta[dd];
                const int ks = (n + M - 1) / M;
                PLAN_REQUIRE(n >= 1 && (mixed ? lane0 + ks <= Lk : ks == k), "class %d tile %zu: surface of %d nodes, %d lanes of %d", c, t, n, k, M);
                if (!mixed) lane0 = g * k;
                if (ft.k & 0x100) PLAN_REQUIRE(n == ks * M, "class %d tile %zu marked full", c, t);
                if (mixed) {
                    const uint8_t *tab = &p.cls[ft.node_base + (int64_t)M * Lk];
                    for (int seg = 0; seg < ks; seg++) {
                        const int e = tab[2 * (lane0 + seg)] | (tab[2 * (lane0 + seg) + 1] << 8);
                        PLAN_REQUIRE((e & 63) == g && ((e >> 6) & 63) == seg && ((e & kLaneLastBit) != 0) == (seg == ks - 1),
                                     "class %d tile %zu: lane table entry of lane %d", c, t, lane0 + seg);
                    }
                }
                if (kFastPAL[c]) {
                    PLAN_REQUIRE(!p.cls.empty(), "palette class without class bytes");
                    for (int i = 0; i < n; i++) {
                        const int lane = lane0 + i / M, j = i % M;
                        const uint8_t cb = p.cls[ft.node_base + (int64_t)lane * M + j];
                        const int vi = cb & (kPalV - 1), ui = (cb >> kPalUShift) & (kPalU - 1);
                        PLAN_REQUIRE((vi < p.pal_ubase || vi >= kPalVMark1) && ui < p.pal_stride - p.pal_ubase && cb < 128 &&
                                     (vi != kPalVMark1 + 1 || j + 1 < M), "class byte %d", (int)cb);
                    }
                }
                if (mixed) lane0 += ks;
            }
            PLAN_REQUIRE(!mixed || lane0 == Lk, "class %d tile %zu: %d lanes used, %d declared", c, t, lane0, Lk);
        }
    }
    PLAN_REQUIRE(p.n_small_plain_tiles <= p.n_small_tiles && p.n_small_tiles <= (int)p.gen_tiles.size(), "small tile counts");
    for (size_t t = 0; t < p.gen_tiles.size(); t++) {
        const GeneralTile &gt = p.gen_tiles[t];
        PLAN_REQUIRE(gt.G >= 1 && gt.G <= kWave && gt.n_max >= 1, "general tile %zu: G %d n_max %d", t, gt.G, gt.n_max);
        PLAN_REQUIRE(gt.node_base >= p.gen_base && gt.node_base + (int64_t)gt.n_max * kWave <= p.node_slots, "general tile %zu: node range", t);
        PLAN_REQUIRE(gt.scratch_base >= 0 && gt.scratch_base + (int64_t)kScratchArrays * gt.n_max * kWave <= p.scratch_slots,
                     "general tile %zu: scratch range", t);
        ranges.push_back({gt.node_base, gt.node_base + (int64_t)gt.n_max * kWave});
        PLAN_REQUIRE(gt.surf_base >= 0 && gt.surf_base + gt.G <= S, "general tile %zu: surfaces", t);
        for (int g = 0; g < gt.G; g++) {
            const int64_t dd = gt.surf_base + g;
            PLAN_REQUIRE(!covered[dd], "device surface %lld in two tiles", (long long)dd);
            covered[dd] = 1;
            PLAN_REQUIRE(p.meta[dd] <= gt.n_max, "general tile %zu: surface longer than n_max", t);
            if ((int)t < p.n_small_tiles) PLAN_REQUIRE(p.meta[dd] <= 4, "small tile %zu holds a surface of %d nodes", t, p.meta[dd]);
        }
    }
    for (int64_t dd = 0; dd < S; dd++) PLAN_REQUIRE(covered[dd], "device surface %lld in no tile", (long long)dd);
    std::sort(ranges.begin(), ranges.end());
    for (size_t i = 1; i < ranges.size(); i++)
        PLAN_REQUIRE(ranges[i].first >= ranges[i - 1].second, "node ranges of two tiles overlap at slot %lld", (long long)ranges[i].first);
    for (int64_t dd = 0; dd < S; dd++) {
        const int64_t s = p.orig_of[dd];
        const int n = (int)(d->node_offset[s + 1] - d->node_offset[s]);
        PLAN_REQUIRE(p.meta[dd] == n && (p.side[dd].kind_n >> 16) == n && (p.side[S + dd].kind_n >> 16) == n, "node count of device surface %lld", (long long)dd);
        PLAN_REQUIRE((p.side[dd].kind_n & 3) == d->front_kind[s] && (p.side[S + dd].kind_n & 3) == d->back_kind[s], "boundary kinds of device surface %lld", (long long)dd);
        PLAN_REQUIRE(((p.side[dd].kind_n >> 4) & 3) == d->back_kind[s] && ((p.side[S + dd].kind_n >> 4) & 3) == d->front_kind[s], "peer kinds of device surface %lld", (long long)dd);
        PLAN_REQUIRE(p.side[dd].zone >= 0 && p.side[S + dd].zone >= 0 && (p.n_zones == 0 || (p.side[dd].zone < p.n_zones && p.side[S + dd].zone < p.n_zones)), "zone fields of device surface %lld (read unconditionally)", (long long)dd);
    }
    // zone contribution lists: one entry per Space-facing side, each side knows its place
    PLAN_REQUIRE((int64_t)p.zone_off.size() == Z + 1 && p.zone_off[0] == 0, "zone offsets");
    for (int64_t z = 0; z < Z; z++) PLAN_REQUIRE(p.zone_off[z + 1] >= p.zone_off[z], "zone offsets not monotone at %lld", (long long)z);
    PLAN_REQUIRE((int64_t)p.zone_entries.size() == (Z > 0 ? p.zone_off[Z] : 0), "zone entries size");
    int64_t n_space_sides = 0;
    for (int64_t rec = 0; rec < 2 * S; rec++) {
        const SideConst &c = p.side[rec];
        if ((c.kind_n & 3) != KIND_SPACE) continue;
        n_space_sides++;
        int64_t pos;
        memcpy(&pos, &c.ambient, sizeof pos);
        PLAN_REQUIRE(c.zone >= 0 && c.zone < Z, "side record %lld: zone", (long long)rec);
        PLAN_REQUIRE(pos >= p.zone_off[c.zone] && pos < p.zone_off[c.zone + 1], "side record %lld: entry position %lld outside its zone's list", (long long)rec, (long long)pos);
        PLAN_REQUIRE((int64_t)p.zone_entries[pos].hs_index == rec, "side record %lld: entry %lld belongs to record %u", (long long)rec, (long long)pos, p.zone_entries[pos].hs_index);
        PLAN_REQUIRE((int64_t)p.zone_entries[pos].t_index < p.node_slots, "entry %lld: face node index", (long long)pos);
    }
    PLAN_REQUIRE(n_space_sides == (int64_t)p.zone_entries.size(), "%lld Space-facing sides, %zu entries", (long long)n_space_sides, p.zone_entries.size());
    // cluster-resident march: workgroups inside the kernel's limits
    PLAN_REQUIRE((int64_t)p.zone_block.size() == Z, "zone_block size");
    PLAN_REQUIRE(p.fzone_eoff.size() == p.fzones.size() + 1, "fused zone offsets");
    int64_t fused_surfaces = 0;
    std::vector<uint8_t> zone_seen(Z, 0);
    for (int c = 0; c < kNumFast; c++)
        for (int g2 = 0; g2 < 4; g2++)
            for (const FusedBlock &fb : p.fblocks[c][g2]) {
                const int fw = (g2 & 1) ? 8 : 4;
                PLAN_REQUIRE(fb.n_tiles >= 0 && fb.n_small >= 0 && fb.n_tiles + fb.n_small >= 1 && fb.n_tiles + fb.n_small <= fw,
                             "class %d list %d: workgroup of %d + %d wavefronts", c, g2, fb.n_tiles, fb.n_small);
                PLAN_REQUIRE((g2 >> 1) || fb.n_small == 0, "plain workgroup with small-surface tiles");
                PLAN_REQUIRE(fb.first_tile >= p.n_stream_tiles[c] && fb.first_tile + fb.n_tiles <= (int)p.fast_tiles[c].size(), "class %d list %d: tile range", c, g2);
                PLAN_REQUIRE(fb.n_small == 0 || (fb.first_small >= p.n_small_plain_tiles + p.n_smallcav_stream_tiles &&
                                                 fb.first_small + fb.n_small <= p.n_small_tiles), "class %d list %d: small tile range", c, g2);
                PLAN_REQUIRE(fb.n_zones >= 0 && fb.n_zones <= kFusedMaxZones && fb.first_zone >= 0 &&
                             fb.first_zone + fb.n_zones <= (int)p.fzones.size(), "class %d list %d: zone range", c, g2);
                const int e0 = p.fzone_eoff[fb.first_zone], e1 = p.fzone_eoff[fb.first_zone + fb.n_zones];
                PLAN_REQUIRE(e1 - e0 <= kFusedMaxEntries && e1 <= (int)p.fslots.size(), "class %d list %d: %d zone-facing sides", c, g2, e1 - e0);
                for (int e = e0; e < e1; e++) PLAN_REQUIRE(p.fslots[e] < 2 * kWave * fw, "slot %d outside the workgroup", (int)p.fslots[e]);
                for (int j = 0; j < fb.n_zones; j++) {
                    const int32_t z = p.fzones[fb.first_zone + j];
                    PLAN_REQUIRE(z >= 0 && z < Z && !zone_seen[z] && p.zone_block[z] >= 0, "fused zone %d", z);
                    zone_seen[z] = 1;
                    // a fused zone's contributions all come from inside the workgroup
                    PLAN_REQUIRE(p.fzone_eoff[fb.first_zone + j + 1] - p.fzone_eoff[fb.first_zone + j] == p.zone_off[z + 1] - p.zone_off[z],
                                 "zone %d: %d sides in the workgroup, %lld in the model", z,
                                 p.fzone_eoff[fb.first_zone + j + 1] - p.fzone_eoff[fb.first_zone + j], (long long)(p.zone_off[z + 1] - p.zone_off[z]));
                }
                for (int q = 0; q < fb.n_tiles; q++) {
                    const FastTile &ft = p.fast_tiles[c][fb.first_tile + q];
                    if (!single_lane_class(c) || (g2 >> 1))
                        for (int g = 0; g < ft.G; g++)
                            PLAN_REQUIRE(p.meta[ft.surf_base + g] > kFastM[c], "fused tile with a single-lane surface (class %d)", c);
                    fused_surfaces += ft.G;
                    for (int g = 0; g < ft.G; g++)
                        for (int side = 0; side < 2; side++) {
                            const int64_t rec = (int64_t)side * S + ft.surf_base + g;
                            const int lz = p.side_lzone[rec];
                            PLAN_REQUIRE(((p.side[rec].kind_n & 3) == KIND_SPACE) == (lz >= 0) && lz < fb.n_zones, "side %lld: local zone %d", (long long)rec, lz);
                            if (lz >= 0) PLAN_REQUIRE(p.fzones[fb.first_zone + lz] == p.side[rec].zone, "side %lld: local zone maps to another zone", (long long)rec);
                        }
                }
                for (int q = 0; q < fb.n_small; q++) fused_surfaces += p.gen_tiles[fb.first_small + q].G;
            }
    // teams (layout.hpp, FusedSuper): members of four wavefronts, a zone's sides spread over the members that list it
    PLAN_REQUIRE(p.team_zinfo.size() == p.fzones.size(), "team zone table size");
    {
        std::vector<int64_t> team_sides(Z, 0);
        for (int c = 0; c < kNumFast; c++) {
            PLAN_REQUIRE(p.team_supers[c].empty() || (kFastPAL[c] && !kFastCAV[c]), "class %d cannot march in teams", c);
            size_t covered = 0;
            for (const FusedSuper &su : p.team_supers[c]) {
                PLAN_REQUIRE(su.n_members >= 2 && su.n_members <= kTeamMax && su.first_block == (int32_t)covered &&
                             su.first_block + su.n_members <= (int32_t)p.team_blocks[c].size(), "class %d: team of %d members at %d", c, su.n_members, su.first_block);
                covered += (size_t)su.n_members;
                std::vector<uint32_t> mask_seen(kTeamZones, 0), mask_told(kTeamZones, 0);
                std::vector<int32_t> slot_zone(kTeamZones, -1);
                for (int m = 0; m < su.n_members; m++) {
                    const FusedBlock &fb = p.team_blocks[c][su.first_block + m];
                    PLAN_REQUIRE(fb.n_tiles >= 1 && fb.n_tiles <= 4 && fb.n_small == 0, "team member of %d + %d wavefronts", fb.n_tiles, fb.n_small);
                    PLAN_REQUIRE(fb.first_tile >= p.n_stream_tiles[c] && fb.first_tile + fb.n_tiles <= (int)p.fast_tiles[c].size(), "team member: tile range");
                    PLAN_REQUIRE(fb.n_zones >= 0 && fb.n_zones <= kFusedMaxZones && fb.first_zone >= 0 &&
                                 fb.first_zone + fb.n_zones <= (int)p.fzones.size(), "team member: zone range");
                    const int e0 = p.fzone_eoff[fb.first_zone], e1 = p.fzone_eoff[fb.first_zone + fb.n_zones];
                    PLAN_REQUIRE(e1 - e0 <= kFusedMaxEntries && e1 <= (int)p.fslots.size(), "team member: %d zone-facing sides", e1 - e0);
                    for (int e = e0; e < e1; e++) PLAN_REQUIRE(p.fslots[e] < 2 * kWave * 4, "slot %d outside the workgroup", (int)p.fslots[e]);
                    for (int j = 0; j < fb.n_zones; j++) {
                        const int32_t z = p.fzones[fb.first_zone + j];
                        const uint32_t info = p.team_zinfo[fb.first_zone + j], slot = info & 0xffffu;
                        PLAN_REQUIRE(z >= 0 && z < Z && p.zone_block[z] >= 0 && slot < (uint32_t)kTeamZones, "team zone %d", z);
                        PLAN_REQUIRE(slot_zone[slot] < 0 || slot_zone[slot] == z, "exchange slot %u names two zones", slot);
                        PLAN_REQUIRE(((info >> 16) >> m) & 1u, "zone %d: member %d lists it but is not among its members", z, m);
                        slot_zone[slot] = z;
                        mask_seen[slot] |= 1u << m;
                        mask_told[slot] = info >> 16;
                        zone_seen[z] = 1;
                        team_sides[z] += p.fzone_eoff[fb.first_zone + j + 1] - p.fzone_eoff[fb.first_zone + j];
                    }
                    for (int q = 0; q < fb.n_tiles; q++) {
                        const FastTile &ft = p.fast_tiles[c][fb.first_tile + q];
                        fused_surfaces += ft.G;
                        for (int g = 0; g < ft.G; g++)
                            for (int side = 0; side < 2; side++) {
                                const int64_t rec = (int64_t)side * S + ft.surf_base + g;
                                const int lz = p.side_lzone[rec];
                                PLAN_REQUIRE(((p.side[rec].kind_n & 3) == KIND_SPACE) == (lz >= 0) && lz < fb.n_zones, "side %lld: local zone %d", (long long)rec, lz);
                                if (lz >= 0) PLAN_REQUIRE(p.fzones[fb.first_zone + lz] == p.side[rec].zone, "side %lld: local zone maps to another zone", (long long)rec);
                            }
                    }
                }
                // every member awaited for a zone really publishes for it, and nobody else does
                for (int q = 0; q < kTeamZones; q++) PLAN_REQUIRE(mask_seen[q] == mask_told[q], "exchange slot %d: members %x listed, %x awaited", q, mask_seen[q], mask_told[q]);
            }
            PLAN_REQUIRE(covered == p.team_blocks[c].size(), "class %d: team members outside any team", c);
        }
        // a zone marched by a team has all its sides inside the team
        for (int64_t z = 0; z < Z; z++)
            if (team_sides[z] > 0) PLAN_REQUIRE(team_sides[z] == p.zone_off[z + 1] - p.zone_off[z], "zone %lld: %lld sides in its team, %lld in the model", (long long)z, (long long)team_sides[z], (long long)(p.zone_off[z + 1] - p.zone_off[z]));
    }
    for (int64_t z = 0; z < Z; z++) PLAN_REQUIRE((p.zone_block[z] >= 0) == (zone_seen[z] != 0), "zone %lld: block table and workgroup lists disagree", (long long)z);
    PLAN_REQUIRE(fused_surfaces == p.n_fused_surfaces, "%lld surfaces in workgroups, %lld counted", (long long)fused_surfaces, (long long)p.n_fused_surfaces);
    if (!p.cavref.empty()) {
        PLAN_REQUIRE((int64_t)p.cavref.size() == 4 * S, "cavref size");
        for (int64_t dd = 0; dd < S; dd++)
            for (int r = 0; r < 2; r++) {
                const int node = p.cavref[4 * dd + 2 * r], cav = p.cavref[4 * dd + 2 * r + 1];
                PLAN_REQUIRE((node < 0) == (cav < 0) && node < p.meta[dd] - 1 + (node < 0) && cav < p.n_cav, "cavity reference of device surface %lld", (long long)dd);
            }
    }
    for (int32_t cidx : p.cav_idx) PLAN_REQUIRE(cidx >= -1 && cidx < p.n_cav, "cavity index %d", cidx);
#undef PLAN_REQUIRE
    return HEAT_OK;
}

// ---------------------------------------------------------------------------
// Partition of a model over ranks along its clusters (include/heat_amd.h, heat_partition).
int partition_surfaces(const heat_batch_desc *d, int32_t n_ranks, int32_t *rank_of_surface, int64_t *n_shared_zones,
                       std::string &err) {
    if (!d || !rank_of_surface) return failp(err, HEAT_E_INVALID_ARG, "NULL argument");
    if (n_ranks < 1) return failp(err, HEAT_E_INVALID_ARG, "n_ranks < 1");
    int rc = check_desc(d, err);
    if (rc) return rc;
    const int64_t S = d->n_surfaces, Z = d->n_zones;
    std::vector<int32_t> cs, cz;
    int32_t nc = 0;
    find_clusters(d, cs, cz, nc);
    auto weight = [&](int64_t s) { return 32.0 * (double)(d->node_offset[s + 1] - d->node_offset[s]) + 152.0; };
    std::vector<double> cw(nc, 0.0);
    double total = 0.0;
    for (int64_t s = 0; s < S; s++) {
        const double w = weight(s);
        total += w;
        if (cs[s] >= 0) cw[cs[s]] += w;
    }
    const double target = total / n_ranks;
    // Items in model order: a whole cluster at the place of its first surface (its later surfaces follow it to its
    // rank), or — a cluster heavier than a quarter of a shard, and surfaces that face no zone — single surfaces.
    std::vector<int32_t> crank(nc, -1);
    double cum = 0.0;
    int32_t r = 0;
    auto advance = [&](double w) {
        // the item goes to the rank its midpoint falls into
        const double mid = cum + 0.5 * w;
        while (r + 1 < n_ranks && mid >= (r + 1) * target) r++;
        cum += w;
        return r;
    };
    for (int64_t s = 0; s < S; s++) {
        const int32_t c = cs[s];
        if (c >= 0 && cw[c] <= 0.25 * target) {
            if (crank[c] < 0) crank[c] = advance(cw[c]);
            rank_of_surface[s] = crank[c];
        } else {
            rank_of_surface[s] = advance(weight(s));
        }
    }
    if (n_shared_zones) {
        std::vector<int32_t> first(Z, -1);
        std::vector<uint8_t> shared(Z, 0);
        auto touch = [&](int32_t z, int32_t rk) {
            if (first[z] < 0) first[z] = rk;
            else if (first[z] != rk) shared[z] = 1;
        };
        for (int64_t s = 0; s < S; s++) {
            if (d->front_kind[s] == HEAT_BOUNDARY_SPACE) touch(d->front_zone[s], rank_of_surface[s]);
            if (d->back_kind[s] == HEAT_BOUNDARY_SPACE) touch(d->back_zone[s], rank_of_surface[s]);
        }
        int64_t n = 0;
        for (int64_t z = 0; z < Z; z++) n += shared[z];
        *n_shared_zones = n;
    }
    return HEAT_OK;
}

// The descriptor of one rank's surfaces (zones, cavities and state slots stay global).
void ShardDesc::build(const heat_batch_desc *d, const int32_t *rank_of_surface, int32_t rank) {
    const int64_t S = d->n_surfaces;
    std::vector<int64_t> idx;
    for (int64_t s = 0; s < S; s++)
        if (rank_of_surface[s] == rank) idx.push_back(s);
    const int64_t n = (int64_t)idx.size();
    node_offset.assign(n + 1, 0);
    for (int64_t q = 0; q < n; q++) node_offset[q + 1] = node_offset[q] + (d->node_offset[idx[q] + 1] - d->node_offset[idx[q]]);
    const int64_t N = node_offset[n];
    auto take_nodes = [&](const double *src, std::vector<double> &dst) {
        dst.resize(N);
        for (int64_t q = 0; q < n; q++)
            std::copy(src + d->node_offset[idx[q]], src + d->node_offset[idx[q] + 1], dst.begin() + node_offset[q]);
    };
    take_nodes(d->mass, mass);
    take_nodes(d->uvalue, uvalue);
    take_nodes(d->front_alpha, front_alpha);
    take_nodes(d->back_alpha, back_alpha);
    if (d->seg_cavity) {
        seg_cavity.resize(N);
        for (int64_t q = 0; q < n; q++)
            std::copy(d->seg_cavity + d->node_offset[idx[q]], d->seg_cavity + d->node_offset[idx[q] + 1], seg_cavity.begin() + node_offset[q]);
    }
    const double *fsrc[] = {d->front_ambient, d->back_ambient, d->front_emissivity, d->back_emissivity, d->area, d->perimeter,
                            d->cos_tilt, d->normal_x, d->normal_y, d->wind_modifier, d->front_hs_fix, d->back_hs_fix};
    for (int a = 0; a < 12; a++) {
        f64[a].clear();
        if (!fsrc[a]) continue;
        f64[a].resize(n);
        for (int64_t q = 0; q < n; q++) f64[a][q] = fsrc[a][idx[q]];
    }
    const int32_t *isrc[] = {d->front_kind, d->back_kind, d->front_zone, d->back_zone};
    for (int a = 0; a < 4; a++) {
        i32[a].resize(n);
        for (int64_t q = 0; q < n; q++) i32[a][q] = isrc[a][idx[q]];
    }
    const int64_t *ssrc[] = {d->first_node_slot, d->hs_front_slot, d->hs_back_slot, d->flow_front_slot, d->flow_back_slot,
                             d->solar_front_slot, d->solar_back_slot, d->ir_front_slot, d->ir_back_slot};
    for (int a = 0; a < 9; a++) {
        i64[a].resize(n);
        for (int64_t q = 0; q < n; q++) i64[a][q] = ssrc[a][idx[q]];
    }
    desc = *d;
    desc.n_surfaces = n;
    desc.node_offset = node_offset.data();
    desc.mass = mass.data();
    desc.uvalue = uvalue.data();
    desc.front_alpha = front_alpha.data();
    desc.back_alpha = back_alpha.data();
    desc.seg_cavity = d->seg_cavity ? seg_cavity.data() : nullptr;
    const double **fdst[] = {&desc.front_ambient, &desc.back_ambient, &desc.front_emissivity, &desc.back_emissivity, &desc.area,
                             &desc.perimeter, &desc.cos_tilt, &desc.normal_x, &desc.normal_y, &desc.wind_modifier,
                             &desc.front_hs_fix, &desc.back_hs_fix};
    for (int a = 0; a < 12; a++) *fdst[a] = fsrc[a] ? f64[a].data() : nullptr;
    const int32_t **idst[] = {&desc.front_kind, &desc.back_kind, &desc.front_zone, &desc.back_zone};
    for (int a = 0; a < 4; a++) *idst[a] = i32[a].data();
    const int64_t **sdst[] = {&desc.first_node_slot, &desc.hs_front_slot, &desc.hs_back_slot, &desc.flow_front_slot,
                              &desc.flow_back_slot, &desc.solar_front_slot, &desc.solar_back_slot, &desc.ir_front_slot,
                              &desc.ir_back_slot};
    for (int a = 0; a < 9; a++) *sdst[a] = i64[a].data();
    original_index = std::move(idx);
}

std::string &last_error() {
    thread_local std::string e;
    return e;
}

}  // namespace heat

// ---------------------------------------------------------------------------
// Host-only entry points of the C ABI (include/heat_amd.h): no device needed.
extern "C" {

const char *heat_last_error(void) { return heat::last_error().c_str(); }
int heat_amd_abi_version(void) { return HEAT_AMD_ABI_VERSION; }

int heat_partition(const heat_batch_desc *desc, int32_t n_ranks, int32_t *rank_of_surface, int64_t *n_shared_zones) {
    return heat::partition_surfaces(desc, n_ranks, rank_of_surface, n_shared_zones, heat::last_error());
}

int heat_plan_check(const heat_batch_desc *desc, const heat_batch_options *opt_in, int64_t summary[8]) {
    heat_batch_options opt;
    memset(&opt, 0, sizeof opt);
    opt.device = -1;
    opt.n_ranks = 1;
    if (opt_in) opt = *opt_in;
    heat::Plan p;
    int rc = heat::make_plan(desc, opt, p, heat::last_error());
    if (rc) return rc;
    rc = heat::check_plan(p, desc, heat::last_error());
    if (rc) return rc;
    if (summary) {
        int64_t n_blocks = 0, n_fast_tiles = 0;
        for (int c = 0; c < heat::kNumFast; c++) {
            n_fast_tiles += (int64_t)p.fast_tiles[c].size();
            for (int g2 = 0; g2 < 4; g2++) n_blocks += (int64_t)p.fblocks[c][g2].size();
            n_blocks += (int64_t)p.team_blocks[c].size();
        }
        summary[0] = p.class_counts[0];
        summary[1] = p.class_counts[1];
        summary[2] = p.class_counts[2];
        summary[3] = p.class_counts[3];
        summary[4] = p.class_counts[4];
        summary[5] = p.n_fused_surfaces;
        summary[6] = n_blocks;
        summary[7] = n_fast_tiles + (int64_t)p.gen_tiles.size();
    }
    return HEAT_OK;
}

}  // extern "C"
