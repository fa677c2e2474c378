// setup.cpp — host-side setup of a thermal model (include/heat_amd_setup.h): the C++ counterpart of
// Discretization::new, Glazing::alphas, the solar-absorption distribution, wind_speed_modifier and the
// timestep selection of ThermalModel::new in the reference. Product code: no dependency on oracle/.
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../include/heat_amd_setup.h"

// The reference (Rust) never fuses a*b+c: keep the setup arithmetic unfused so that the discretization
// (element counts come from a floor()) is the reference's to the last bit.
#pragma clang fp contract(off)

namespace {

constexpr double kMaxRs = 0.05;     // reference src/discretization.rs:21 (MAX_RS)
constexpr double kHsInit = 1.739658084820765;  // src/surface_trait.rs:231,248
constexpr double kTInit = 22.0;     // src/surface_trait.rs:368; src/zone.rs:48
constexpr double kPi = 3.14159265358979323846264338327950288;

// ---- Discretization (src/discretization.rs) --------------------------------------------------------
struct Discretization {
    int tstep_subdivision = 1;
    std::vector<int32_t> n_elements;
    std::vector<double> mass, uvalue;
    std::vector<int32_t> seg_cavity;
    std::vector<heat_cavity> cavities;
};

// discretize_construction, :410-544 — the recursion `aux(.., n + 1, ..)` restarts the layer loop with a
// finer timestep; written here as a loop over n.
// What Discretization::new takes for granted (the reference loops for ever, or panics on an allocation, where these do not
// hold): positive finite timesteps and element size, a refinement that ends within a million subdivisions, positive finite
// thickness / conductivity / density / specific heat of every solid layer. Found by tools/fuzz_setup.py's damaged constructions.
static bool positive(double x) { return x > 0.0 && std::isfinite(x); }
int check_construction(int32_t n_layers, const heat_layer *layers, double model_dt, double max_dx, double min_dt) {
    if (!positive(model_dt) || !positive(max_dx) || !positive(min_dt) || model_dt / min_dt > 1e6) return HEAT_E_INVALID_ARG;
    for (int32_t l = 0; l < n_layers; l++) {
        const heat_layer &L = layers[l];
        if (!positive(L.thickness)) return HEAT_E_INVALID_ARG;
        if (!L.is_gas && (!positive(L.conductivity) || !positive(L.density) || !positive(L.specific_heat))) return HEAT_E_INVALID_ARG;
    }
    return HEAT_OK;
}

int discretize(int32_t n_layers, const heat_layer *layers, double model_dt, double max_dx, double min_dt,
               int32_t *n_elements) {
    if (int rc = check_construction(n_layers, layers, model_dt, max_dx, min_dt)) return rc;
    for (int n = 1;; n++) {
        const double dt = model_dt / static_cast<double>(n);
        const bool can_refine = model_dt / static_cast<double>(n + 1) > min_dt;
        bool restart = false;
        for (int32_t l = 0; l < n_layers && !restart; l++) {
            const heat_layer &L = layers[l];
            if (L.is_gas) {  // :447-450
                n_elements[l] = 0;
                continue;
            }
            // positive root of dx^2 - dt/(rho cp Rs) dx - 2 dt k/(rho cp) = 0, :453-465
            const double a_coef = 1.;
            const double b_coef = -dt / (L.density * L.specific_heat * kMaxRs);
            const double c_coef = -2. * dt * L.conductivity / (L.density * L.specific_heat);
            const double disc = b_coef * b_coef - 4. * a_coef * c_coef;
            const double min_dx = (-b_coef + std::sqrt(disc)) / (2. * a_coef);
            if (min_dx > L.thickness) {  // :467-479
                if (can_refine) restart = true;
                else n_elements[l] = 0;
            } else {  // :480-502
                const double m = std::floor(L.thickness / min_dx);
                if (!(m >= 1.0 && m <= 1e6)) return HEAT_E_TOO_MANY_NODES;  // (a layer of more than a million elements)
                const double dx = L.thickness / m;
                if (dx > max_dx) {
                    if (can_refine) restart = true;
                    else n_elements[l] = 0;
                } else {
                    n_elements[l] = static_cast<int32_t>(m);
                }
            }
        }
        if (!restart) return n;
    }
}

int32_t count_nodes(int32_t n_layers, const int32_t *n_elements) {  // :174-176
    int32_t sum = 0, zeros = 0;
    for (int32_t l = 0; l < n_layers; l++) {
        sum += n_elements[l];
        zeros += (n_elements[l] == 0);
    }
    return sum + zeros + 1;
}

// build, :163-298. Returns the number of cavities or a negative status.
int build_segments(int32_t n_layers, const heat_layer *layers, const int32_t *n_elements, double height,
                   double angle, double *mass, double *uvalue, int32_t *seg_cavity, heat_cavity *cavities,
                   int32_t cav_base) {
    const int32_t n_nodes = count_nodes(n_layers, n_elements);
    for (int32_t i = 0; i < n_nodes; i++) {
        mass[i] = 0.0;
        uvalue[i] = std::numeric_limits<double>::quiet_NaN();  // UValue::None
        seg_cavity[i] = -1;
    }
    int32_t segment = 0, n_cav = 0;
    for (int32_t l = 0; l < n_layers; l++) {
        const heat_layer &L = layers[l];
        int32_t n = n_elements[l];
        double layer_mass = 0.0;  // :190-202
        if (n != 0 && !L.is_gas) layer_mass = L.density * L.specific_heat * (L.thickness / static_cast<double>(n));
        if (n == 0) n = 1;
        for (int32_t e = 0; e < n; e++, segment++) {
            if (!L.is_gas) {  // :210-220
                mass[segment] += layer_mass / 2.;
                mass[segment + 1] += layer_mass / 2.;
                uvalue[segment] = L.conductivity / (L.thickness / static_cast<double>(n));
            } else {  // :221-285
                if (l == 0) return HEAT_E_INVALID_ARG;             // gas as the first layer
                if (l + 1 >= n_layers) return HEAT_E_INVALID_ARG;  // gas as the last layer
                if (layers[l - 1].is_gas || layers[l + 1].is_gas) return HEAT_E_INVALID_ARG;  // two gases together
                heat_cavity &c = cavities[n_cav];
                c.thickness = L.thickness;
                c.height = height;
                c.angle = angle;
                c.eout = layers[l - 1].back_thermal_absorbtance;
                c.ein = layers[l + 1].front_thermal_absorbtance;
                c.gas = L.gas;
                c.reserved = 0;
                uvalue[segment] = 0.0;
                seg_cavity[segment] = cav_base + n_cav;
                n_cav++;
            }
        }
    }
    uvalue[n_nodes - 1] = 0.0;  // UValue::Back, :290
    seg_cavity[n_nodes - 1] = -1;
    return n_cav;
}

// ---- Glazing (src/glazing.rs) ------------------------------------------------------------------------
struct Glazing {
    double tau, rho_front, rho_back, alpha_front, alpha_back;
    Glazing(double t, double rf, double rb)  // :50-65
        : tau(t), rho_front(rf), rho_back(rb), alpha_front(1. - t - rf), alpha_back(1. - t - rb) {}
    Glazing combine(const Glazing &o) const {  // :180-221 (ISO 9050 eqs. 2, 5)
        const double rb = o.rho_back + o.tau * o.tau * rho_back / (1. - o.rho_front * rho_back);
        const double rf = rho_front + tau * tau * o.rho_front / (1. - rho_back * o.rho_front);
        const double t = tau * o.tau / (1. - rho_back * o.rho_front);
        return Glazing(t, rf, rb);
    }
    static Glazing combine_layers(const Glazing *layers, int n) {  // :223-233
        if (n == 1) return layers[0];
        return layers[0].combine(combine_layers(layers + 1, n - 1));
    }
    // absorbed in self (.first) and in other (.second), ISO 9050 eqs. 17-18, :247-252
    std::pair<double, double> combined_alphas(const Glazing &o) const {
        const double denom = 1. - rho_back * o.rho_front;
        return {alpha_front + alpha_back * tau * o.rho_front / denom, o.alpha_front * tau / denom};
    }
    static std::vector<double> alphas(const std::vector<Glazing> &layers) {  // :259-286
        std::vector<double> ret;
        const int n = static_cast<int>(layers.size());
        if (n == 0) return ret;
        if (n == 1) return {layers[0].alpha_front};
        double acc = 0.0;
        for (int i = 1; i < n; i++) {
            const Glazing g0 = combine_layers(layers.data(), i);
            const Glazing g1 = combine_layers(layers.data() + i, n - i);
            const double a0 = g0.combined_alphas(g1).first;
            ret.push_back(a0 - acc);
            acc = a0;
        }
        const Glazing g0 = combine_layers(layers.data(), n - 1);
        ret.push_back(g0.combined_alphas(layers[n - 1]).second);
        return ret;
    }
};

// get_front/back_glazing_system -> get_glazing_from_iter, :67-142. Negative where the reference panics.
int glazing_system(int32_t n_layers, const heat_layer *layers, bool from_back, std::vector<Glazing> &out) {
    out.clear();
    int32_t pos = 0;
    auto at = [&](int32_t p) -> const heat_layer & { return layers[from_back ? n_layers - 1 - p : p]; };
    for (;;) {
        if (pos >= n_layers) return HEAT_E_INVALID_ARG;  // i.next().unwrap() on an empty iterator
        const heat_layer &L = at(pos++);
        if (L.is_gas) return HEAT_E_INVALID_ARG;  // "NOT expecting a gas"
        out.emplace_back(L.solar_transmittance, 1. - L.solar_transmittance - L.front_solar_absorbtance,
                         1. - L.solar_transmittance - L.back_solar_absorbtance);
        if (L.solar_transmittance < 1e-9) break;  // opaque: nothing behind it matters
        if (pos >= n_layers) break;
        if (!at(pos++).is_gas) return HEAT_E_INVALID_ARG;  // "Expecting a Gas"
    }
    return static_cast<int>(out.size());
}

// src/surface.rs:463-537
int node_alphas(int32_t n_layers, const heat_layer *layers, const int32_t *n_elements, int32_t n_nodes,
                double *front_alphas, double *back_alphas) {
    std::vector<Glazing> sys;
    int rc = glazing_system(n_layers, layers, false, sys);
    if (rc < 0) return rc;
    std::vector<double> prev = Glazing::alphas(sys);
    if (prev.size() != 1 && static_cast<int32_t>(prev.size()) != n_layers) return HEAT_E_INVALID_ARG;  // :470-472
    for (int32_t i = 0; i < n_nodes; i++) front_alphas[i] = back_alphas[i] = 0.0;
    int32_t global_i = 0;
    for (size_t ai = 0; ai < prev.size(); ai++) {  // :478-503
        const int32_t layer = 2 * static_cast<int32_t>(ai);  // skip cavities
        const int32_t n = n_elements[layer] == 0 ? 1 : n_elements[layer];
        if (layers[layer].is_gas) return HEAT_E_INVALID_ARG;
        if (layers[layer].solar_transmittance > 0.0) {
            for (int32_t li = 0; li <= n; li++) front_alphas[global_i + li] += prev[ai] / static_cast<double>(n + 1);
        } else {
            front_alphas[global_i] += prev[ai];
        }
        global_i += n + 1;
    }
    rc = glazing_system(n_layers, layers, true, sys);
    if (rc < 0) return rc;
    prev = Glazing::alphas(sys);
    if (prev.size() != 1 && static_cast<int32_t>(prev.size()) != n_layers) return HEAT_E_INVALID_ARG;  // :506-508
    global_i = n_nodes;
    for (size_t ai = 0; ai < prev.size(); ai++) {  // :511-537
        const int32_t layer = n_layers - 2 * static_cast<int32_t>(ai) - 1;
        const int32_t n = n_elements[layer] == 0 ? 1 : n_elements[layer];
        if (layers[layer].is_gas) return HEAT_E_INVALID_ARG;
        if (layers[layer].solar_transmittance > 0.0) {
            for (int32_t li = 0; li <= n; li++) back_alphas[global_i - li - 1] += prev[ai] / static_cast<double>(n + 1);
        } else {
            back_alphas[global_i - 1] += prev[ai];
        }
        global_i -= n + 1;
    }
    return HEAT_OK;
}

double wind_modifier(double height, int32_t terrain) {  // src/surface.rs:135-166
    if (height < 1e-5) return 0.0;
    double alpha = 0.0, delta = 0.0;
    switch (terrain) {
    case HEAT_TERRAIN_COUNTRY: alpha = 0.14; delta = 270.; break;
    case HEAT_TERRAIN_SUBURBS: alpha = 0.22; delta = 370.; break;
    case HEAT_TERRAIN_CITY: alpha = 0.33; delta = 460.; break;
    case HEAT_TERRAIN_OCEAN: alpha = 0.10; delta = 210.; break;
    case HEAT_TERRAIN_URBAN: alpha = 0.22; delta = 370.; break;
    case HEAT_TERRAIN_NO_SITE_DETAILS: alpha = 0.22; delta = 370.; break;  // defaults to Urban
    default: break;  // Some(details) without terrain
    }
    return std::pow(270. / 10., 0.14) * std::pow(height / delta, alpha);
}

}  // namespace

// ---- ThermalModel::new as a builder ---------------------------------------------------------------------
struct heat_model_builder {
    int32_t n_per_hour = 1;
    int32_t terrain = HEAT_TERRAIN_NO_SITE_DETAILS;
    struct SurfaceIn {
        heat_surface_in in;
        std::vector<heat_layer> layers;
        Discretization d;
        std::vector<double> front_alpha, back_alpha;
    };
    std::vector<SurfaceIn> surfaces;
    std::vector<double> zone_volume;
    bool finished = false;

    // flattened output
    heat_batch_desc desc;
    std::vector<int64_t> node_offset, first_node_slot, slots[8], zone_slot;
    std::vector<double> mass, uvalue, front_alpha, back_alpha, f64[10], state;
    std::vector<int32_t> seg_cavity, i32[4];
    std::vector<heat_cavity> cavities;
    std::vector<size_t> order;  // surfaces first, then fenestrations
    int32_t dt_subdivisions = 1;
};

extern "C" {

int heat_discretize_construction(int32_t n_layers, const heat_layer *layers, double model_dt, double max_dx,
                                 double min_dt, int32_t *n_elements) {
    if (n_layers <= 0 || !layers || !n_elements) return HEAT_E_INVALID_ARG;
    return discretize(n_layers, layers, model_dt, max_dx, min_dt, n_elements);
}

int32_t heat_count_nodes(int32_t n_layers, const int32_t *n_elements) { return count_nodes(n_layers, n_elements); }

int heat_build_segments(int32_t n_layers, const heat_layer *layers, const int32_t *n_elements, double height,
                        double angle, double *mass, double *uvalue, int32_t *seg_cavity, heat_cavity *cavities,
                        int32_t cav_base) {
    if (n_layers <= 0 || !layers || !n_elements || !mass || !uvalue || !seg_cavity) return HEAT_E_INVALID_ARG;
    return build_segments(n_layers, layers, n_elements, height, angle, mass, uvalue, seg_cavity, cavities, cav_base);
}

int heat_get_chunks(int32_t n_nodes, const double *mass, int32_t *n_massive, int32_t *massive_chunks,
                    int32_t *n_nomass, int32_t *nomass_chunks) {  // src/discretization.rs:117-160
    if (n_nodes < 0 || !mass || !n_massive || !n_nomass) return HEAT_E_INVALID_ARG;
    *n_massive = *n_nomass = 0;
    int32_t i = 0;
    while (i < n_nodes) {
        const bool massive = mass[i] >= 1e-5;
        const int32_t ini = i;
        while (i < n_nodes && (mass[i] >= 1e-5) == massive) i++;
        int32_t *dst = massive ? massive_chunks : nomass_chunks;
        int32_t *cnt = massive ? n_massive : n_nomass;
        dst[2 * *cnt] = ini;
        dst[2 * *cnt + 1] = i;
        (*cnt)++;
    }
    return HEAT_OK;
}

int heat_glazing_alphas(int32_t n, const double *tau, const double *rho_front, const double *rho_back,
                        double *alphas) {
    if (n < 0 || !alphas) return HEAT_E_INVALID_ARG;
    std::vector<Glazing> g;
    for (int32_t i = 0; i < n; i++) g.emplace_back(tau[i], rho_front[i], rho_back[i]);
    const std::vector<double> a = Glazing::alphas(g);
    for (size_t i = 0; i < a.size(); i++) alphas[i] = a[i];
    return static_cast<int>(a.size());
}

int heat_node_alphas(int32_t n_layers, const heat_layer *layers, const int32_t *n_elements, int32_t n_nodes,
                     double *front_alphas, double *back_alphas) {
    if (n_layers <= 0 || !layers || !n_elements || !front_alphas || !back_alphas) return HEAT_E_INVALID_ARG;
    return node_alphas(n_layers, layers, n_elements, n_nodes, front_alphas, back_alphas);
}

double heat_wind_speed_modifier(double height, int32_t terrain) { return wind_modifier(height, terrain); }

heat_model_builder *heat_model_builder_create(int32_t n_per_hour, int32_t terrain) {
    if (n_per_hour <= 0) return nullptr;
    heat_model_builder *mb = new heat_model_builder();
    mb->n_per_hour = n_per_hour;
    mb->terrain = terrain;
    return mb;
}

void heat_model_builder_destroy(heat_model_builder *mb) { delete mb; }

int heat_model_builder_add_zone(heat_model_builder *mb, double volume) {
    if (!mb || mb->finished) return HEAT_E_INVALID_ARG;
    mb->zone_volume.push_back(volume);  // ThermalZone::from_space, src/zone.rs:38-56
    return static_cast<int>(mb->zone_volume.size()) - 1;
}

int heat_model_builder_add_surface(heat_model_builder *mb, const heat_surface_in *s) {
    if (!mb || mb->finished || !s || s->n_layers <= 0 || !s->layers) return HEAT_E_INVALID_ARG;
    heat_model_builder::SurfaceIn si;
    si.in = *s;
    si.layers.assign(s->layers, s->layers + s->n_layers);
    si.in.layers = nullptr;
    mb->surfaces.push_back(std::move(si));
    return static_cast<int>(mb->surfaces.size()) - 1;
}

int heat_model_builder_finish(heat_model_builder *mb, const heat_batch_desc **desc_out, const double **state_out,
                              int32_t *dt_subdivisions_out) {
    if (!mb || !desc_out) return HEAT_E_INVALID_ARG;
    if (!mb->finished) {
        const double main_dt = 60. * 60. / static_cast<double>(mb->n_per_hour);  // model.rs:240
        const double max_dx = 0.04, min_dt = 60.;                                // model.rs:236-237
        int32_t subdivisions = 1;
        for (auto &s : mb->surfaces) {
            const int32_t nl = static_cast<int32_t>(s.layers.size());
            const double nz = s.in.normal[2];  // cos_tilt = normal * (0,0,1), model.rs:249
            const double height = 1.;          // model.rs:252
            const double angle = std::acos(nz);  // model.rs:253
            Discretization &d = s.d;
            d.n_elements.resize(nl);
            d.tstep_subdivision = discretize(nl, s.layers.data(), main_dt, max_dx, min_dt, d.n_elements.data());
            if (d.tstep_subdivision < 0) return d.tstep_subdivision;
            const int32_t nn = count_nodes(nl, d.n_elements.data());
            d.mass.resize(nn);
            d.uvalue.resize(nn);
            d.seg_cavity.resize(nn);
            d.cavities.resize(nl);
            const int nc = build_segments(nl, s.layers.data(), d.n_elements.data(), height, angle, d.mass.data(),
                                          d.uvalue.data(), d.seg_cavity.data(), d.cavities.data(), 0);
            if (nc < 0) return nc;
            d.cavities.resize(nc);
            s.front_alpha.resize(nn);
            s.back_alpha.resize(nn);
            const int rc = node_alphas(nl, s.layers.data(), d.n_elements.data(), nn, s.front_alpha.data(),
                                       s.back_alpha.data());
            if (rc < 0) return rc;
            if (d.tstep_subdivision > subdivisions) subdivisions = d.tstep_subdivision;  // model.rs:261-263
        }
        // model.rs:326-331
        double dt = 60. * 60. / (static_cast<double>(mb->n_per_hour) * static_cast<double>(subdivisions));
        const int SAFETY = 2;
        dt /= static_cast<double>(SAFETY);
        mb->dt_subdivisions = subdivisions * SAFETY;

        // surfaces, then fenestrations (model.rs:244-323)
        for (size_t i = 0; i < mb->surfaces.size(); i++) if (!mb->surfaces[i].in.is_fenestration) mb->order.push_back(i);
        for (size_t i = 0; i < mb->surfaces.size(); i++) if (mb->surfaces[i].in.is_fenestration) mb->order.push_back(i);

        const int64_t S = static_cast<int64_t>(mb->order.size()), Z = static_cast<int64_t>(mb->zone_volume.size());
        mb->node_offset.assign(1, 0);
        mb->zone_slot.resize(Z);
        mb->state.clear();
        for (int64_t z = 0; z < Z; z++) {  // zone dry-bulb slots first (model.rs:225-230; zone.rs:45-50)
            mb->zone_slot[z] = static_cast<int64_t>(mb->state.size());
            mb->state.push_back(kTInit);
        }
        for (int64_t q = 0; q < S; q++) {
            const auto &s = mb->surfaces[mb->order[q]];
            const int32_t nn = static_cast<int32_t>(s.d.mass.size());
            const int32_t cav_base = static_cast<int32_t>(mb->cavities.size());
            for (int32_t i = 0; i < nn; i++) {
                mb->mass.push_back(s.d.mass[i]);
                mb->uvalue.push_back(s.d.uvalue[i]);
                mb->seg_cavity.push_back(s.d.seg_cavity[i] >= 0 ? s.d.seg_cavity[i] + cav_base : -1);
                mb->front_alpha.push_back(s.front_alpha[i]);
                mb->back_alpha.push_back(s.back_alpha[i]);
            }
            for (const heat_cavity &c : s.d.cavities) mb->cavities.push_back(c);
            mb->node_offset.push_back(mb->node_offset.back() + nn);
            // state registration order: surface.rs:428-442
            const double init[8] = {kHsInit, kHsInit, 0., 0., 0., 0., 0., 0.};
            for (int a = 0; a < 8; a++) {
                mb->slots[a].push_back(static_cast<int64_t>(mb->state.size()));
                mb->state.push_back(init[a]);
            }
            mb->first_node_slot.push_back(static_cast<int64_t>(mb->state.size()));
            for (int32_t i = 0; i < nn; i++) mb->state.push_back(kTInit);
            mb->i32[0].push_back(s.in.front_kind);
            mb->i32[1].push_back(s.in.back_kind);
            mb->i32[2].push_back(s.in.front_zone);
            mb->i32[3].push_back(s.in.back_zone);
            const double vals[10] = {s.in.front_ambient, s.in.back_ambient,
                                     s.layers.front().front_thermal_absorbtance,   // surface.rs:450-455
                                     s.layers.back().back_thermal_absorbtance,     // surface.rs:456-461
                                     s.in.area, s.in.perimeter, s.in.normal[2], s.in.normal[0], s.in.normal[1],
                                     wind_modifier(s.in.centroid_z, mb->terrain)};
            for (int a = 0; a < 10; a++) mb->f64[a].push_back(vals[a]);
        }
        heat_batch_desc &D = mb->desc;
        std::memset(&D, 0, sizeof D);
        D.abi_version = HEAT_AMD_ABI_VERSION;
        D.n_surfaces = S;
        D.n_zones = Z;
        D.n_cavities = static_cast<int64_t>(mb->cavities.size());
        D.n_state = static_cast<int64_t>(mb->state.size());
        D.dt = dt;
        D.node_offset = mb->node_offset.data();
        D.mass = mb->mass.data();
        D.uvalue = mb->uvalue.data();
        D.seg_cavity = D.n_cavities ? mb->seg_cavity.data() : nullptr;
        D.front_alpha = mb->front_alpha.data();
        D.back_alpha = mb->back_alpha.data();
        D.cavities = D.n_cavities ? mb->cavities.data() : nullptr;
        D.front_kind = mb->i32[0].data(); D.back_kind = mb->i32[1].data();
        D.front_zone = mb->i32[2].data(); D.back_zone = mb->i32[3].data();
        D.front_ambient = mb->f64[0].data(); D.back_ambient = mb->f64[1].data();
        D.front_emissivity = mb->f64[2].data(); D.back_emissivity = mb->f64[3].data();
        D.area = mb->f64[4].data(); D.perimeter = mb->f64[5].data();
        D.cos_tilt = mb->f64[6].data(); D.normal_x = mb->f64[7].data(); D.normal_y = mb->f64[8].data();
        D.wind_modifier = mb->f64[9].data();
        D.first_node_slot = mb->first_node_slot.data();
        D.hs_front_slot = mb->slots[0].data(); D.hs_back_slot = mb->slots[1].data();
        D.flow_front_slot = mb->slots[2].data(); D.flow_back_slot = mb->slots[3].data();
        D.solar_front_slot = mb->slots[4].data(); D.solar_back_slot = mb->slots[5].data();
        D.ir_front_slot = mb->slots[6].data(); D.ir_back_slot = mb->slots[7].data();
        D.zone_volume = mb->zone_volume.data();
        D.zone_slot = mb->zone_slot.data();
        mb->finished = true;
    }
    *desc_out = &mb->desc;
    if (state_out) *state_out = mb->state.data();
    if (dt_subdivisions_out) *dt_subdivisions_out = mb->dt_subdivisions;
    return HEAT_OK;
}

int heat_model_builder_surface_info(const heat_model_builder *mb, int64_t i, int32_t *tstep_subdivision,
                                    int32_t *n_nodes, int32_t *n_elements, int32_t n_elements_cap) {
    if (!mb || !mb->finished || i < 0 || i >= static_cast<int64_t>(mb->order.size())) return HEAT_E_INVALID_ARG;
    const auto &s = mb->surfaces[mb->order[i]];
    if (tstep_subdivision) *tstep_subdivision = s.d.tstep_subdivision;
    if (n_nodes) *n_nodes = static_cast<int32_t>(s.d.mass.size());
    if (n_elements)
        for (int32_t l = 0; l < n_elements_cap && l < static_cast<int32_t>(s.d.n_elements.size()); l++)
            n_elements[l] = s.d.n_elements[l];
    return static_cast<int>(s.d.n_elements.size());
}

}  // extern "C"
