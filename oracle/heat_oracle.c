/*
 * heat_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY (see heat_oracle.h).
 *
 * Plain-C restatement of the reference's wall heat-conduction hot path
 * (SIMPLE-BuildingSimulation/heat v1.0.2). Every function cites the reference
 * lines it follows (paths relative to /root/reference/). Operation order
 * follows the Rust source; tri-diagonal matrices are stored as three diagonals
 * instead of the reference's dense n×n `Matrix` (only the three diagonals are
 * ever non-zero: discretization.rs:642-654, surface.rs:174-183), so this code
 * is FASTER than the real reference — say so wherever it is timed.
 *
 * Build with -ffp-contract=off: Rust never contracts a*b+c into an FMA.
 */
#include "heat_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define OR_PI 3.14159265358979323846264338327950288 /* lib.rs:45 (std::f64::consts::PI) */
#define OR_SIGMA 5.670374419e-8                     /* lib.rs:49 */
#define OR_MAX_NODES 1024

/* f64::powi as lowered through compiler-rt's __powidf2. */
static double or_powi(double a, int b) {
    const int recip = b < 0;
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}

/* f64::to_radians: self * (PI / 180). */
static double or_to_radians(double deg) { return deg * (OR_PI / 180.0); }

/* ------------------------------------------------------------------ */
/* surface.rs:37-46 */
int or_is_windward(double wind_direction, double cos_tilt, double nx, double ny) {
    if (fabs(cos_tilt) < 0.98) {
        double wx = sin(wind_direction), wy = cos(wind_direction);
        /* normal * Vector3D(sin, cos, 0.0) : dot product; the z term is nz*0 */
        return (nx * wx + ny * wy) > 0.0;
    }
    return 1;
}

/* surface.rs:135-166 */
double or_wind_speed_modifier(double height, int has_site_details, int terrain) {
    if (height < 1e-5) return 0.0;
    double alpha = 0.0, delta = 0.0;
    if (has_site_details) {
        switch (terrain) {
        case 0: alpha = 0.14; delta = 270.; break; /* Country */
        case 1: alpha = 0.22; delta = 370.; break; /* Suburbs */
        case 2: alpha = 0.33; delta = 460.; break; /* City */
        case 3: alpha = 0.10; delta = 210.; break; /* Ocean */
        case 4: alpha = 0.22; delta = 370.; break; /* Urban */
        default: break; /* Some(details) without terrain: alpha = delta = 0 */
        }
    } else {
        alpha = 0.22;
        delta = 370.;
    }
    return pow(270. / 10., 0.14) * pow(height / delta, alpha);
}

/* convection.rs:87-110 */
double or_tarp_natural(double air_t, double surf_t, double cos_tilt, int *err) {
    const double MIN_H = 0.1; /* convection.rs:22 */
    double delta_t = air_t - surf_t;
    double abs_delta_t = fabs(delta_t);
    double h;
    if (fabs(delta_t) < 1e-3 || fabs(cos_tilt) < 1e-3) {
        h = 1.31 * pow(abs_delta_t, 1. / 3.);
    } else if ((delta_t < 0. && cos_tilt < 0.) || (delta_t > 0. && cos_tilt > 0.)) {
        h = 9.482 * pow(abs_delta_t, 1. / 3.) / (7.238 - fabs(cos_tilt));
    } else if ((delta_t > 0. && cos_tilt < 0.) || (delta_t < 0. && cos_tilt > 0.)) {
        h = 1.81 * pow(abs_delta_t, 1. / 3.) / (1.382 + fabs(cos_tilt));
    } else {
        if (err) *err = OR_ERR_UNREACHABLE; /* unreachable!() — NaN inputs */
        return NAN;
    }
    return (h < MIN_H) ? MIN_H : h;
}

/* convection.rs:151-168 (roughness_index is always 1 -> COEFFICIENTS[1] = 1.67) */
double or_tarp_total(double air_t, double surf_t, double cos_tilt, double air_speed,
                     double area, double perimeter, int windward, int *err) {
    const double rf = 1.67;
    double wf = windward ? 1.0 : 0.5;
    double forced = 2.537 * wf * rf * sqrt(perimeter * air_speed / area);
    double natural = or_tarp_natural(air_t, surf_t, cos_tilt, err);
    return forced + natural;
}

/* ------------------------------------------------------------------ */
/* gas.rs:45-74: poly![a0, a1] evaluated as a0 + a1*x */
static const double GAS_K[4][2] = {{2.873e-3, 7.760e-5}, {2.285e-3, 5.149e-5}, {9.443e-4, 2.826e-5}, {4.538e-4, 1.723e-5}};
static const double GAS_MU[4][2] = {{3.723e-6, 4.94e-8}, {3.379e-6, 6.451e-8}, {2.213e-6, 7.777e-8}, {1.069e-6, 7.414e-8}};
static const double GAS_CP[4][2] = {{1002.7370, 1.2324e-2}, {521.9285, 0.0}, {248.0907, 0.0}, {158.3397, 0.0}};
static const int GAS_CP_DEG[4] = {1, 0, 0, 0};
static const double GAS_MASS[4] = {28.97, 39.948, 83.8, 131.30};

double or_gas_thermal_conductivity(int gas, double t) { return GAS_K[gas][0] + GAS_K[gas][1] * t; }  /* gas.rs:155-157 */
double or_gas_dynamic_viscosity(int gas, double t) { return GAS_MU[gas][0] + GAS_MU[gas][1] * t; }   /* gas.rs:160-162 */
double or_gas_heat_capacity(int gas, double t) {                                                    /* gas.rs:165-167 */
    return GAS_CP_DEG[gas] ? GAS_CP[gas][0] + GAS_CP[gas][1] * t : GAS_CP[gas][0];
}
double or_gas_mass(int gas) { return GAS_MASS[gas]; } /* gas.rs:170-172 */
/* gas.rs:175-179 */
double or_gas_density(int gas, double temp) {
    const double R = 8314.46261815324;
    return 101325. * GAS_MASS[gas] / (R * temp);
}

static double in_kelvin(double t) { return t + 273.15; } /* gas.rs:183-185 */

/* gas.rs:82-102 */
double or_raleigh(int gas, double t_front, double t_back, double thickness) {
    const double G = 9.81;
    if (fabs(t_front - t_back) < 1e-10) return 0.0000001;
    double temp = (in_kelvin(t_front) + in_kelvin(t_back)) / 2.;
    double beta = 1. / temp;
    double c_p = or_gas_heat_capacity(gas, temp);
    double mu = or_gas_dynamic_viscosity(gas, temp);
    double lambda = or_gas_thermal_conductivity(gas, temp);
    double rho = or_gas_density(gas, temp);
    return or_powi(rho, 2) * or_powi(thickness, 3) * G * beta * c_p * fabs(t_front - t_back) / (mu * lambda);
}

/* gas.rs:285-307 */
static double nu_90(double ra, double a_gi, int *err) {
    double nu1;
    if (ra <= 1e4) {
        nu1 = 1. + 1.7596678 * 1e-10 * pow(ra, 2.2984755);
    } else if (ra < 5e4) {
        nu1 = 0.028154 * pow(ra, 0.4134);
    } else if (ra > 5e4) {
        nu1 = 0.0673838 * pow(ra, 1. / 3.);
    } else {
        if (err) *err = OR_ERR_UNREACHABLE; /* ra == 5e4 exactly, or NaN */
        return NAN;
    }
    double nu2 = 0.242 * pow(ra / a_gi, 0.272);
    return (nu1 > nu2) ? nu1 : nu2;
}

/* gas.rs:249-263 */
static double nu_60(double ra, double a_gi) {
    double g = 0.5 / pow(1. + pow(ra / 3160., 20.6), 0.1);
    double nu1 = pow(1. + or_powi(0.0936 * pow(ra, 0.314) / (1. + g), 7), 1. / 7.);
    double nu2 = (0.104 + 0.175 / a_gi) * pow(ra, 0.283);
    return (nu1 > nu2) ? nu1 : nu2;
}

/* gas.rs:227-244 */
static double nu_0_60(double ra, double gamma) {
    double cos_gamma = cos(gamma);
    double x;
    x = 1. - 1708. / (ra * cos_gamma);
    double a = (x + fabs(x)) / 2.;
    double b = 1. - 1708. * pow(sin(1.8 * gamma), 1.6) / (ra * cos_gamma);
    double c = pow(ra * cos_gamma / 5830., 1. / 3.) - 1.;
    return 1. + 1.44 * a * b + (c + fabs(c)) / 2.;
}

/* gas.rs:269-280 */
static double nu_60_90(double ra, double gamma, double a_gi, int *err) {
    double nu60 = nu_60(ra, a_gi);
    double nu90 = nu_90(ra, a_gi, err);
    double x = (gamma - OR_PI / 3.) / (OR_PI / 2. - OR_PI / 3.);
    return nu60 + (nu90 - nu60) * x;
}

/* gas.rs:312-315 */
static double nu_90_180(double ra, double a_gi, double gamma, int *err) {
    double nu_v = nu_90(ra, a_gi, err);
    return 1. + (nu_v - 1.) * sin(gamma);
}

/* gas.rs:197-221 */
double or_nusselt(double ra, double gamma, double a_gi, int *err) {
    const double THIRTY_RAD = 30. * OR_PI / 180.;
    const double EPSILON_RAD = 0.5 * OR_PI / 180.;
    gamma = fmod(gamma, OR_PI);
    if (gamma >= 0.0 && gamma < 2. * THIRTY_RAD - EPSILON_RAD) {
        return nu_0_60(ra, gamma);
    } else if (gamma < 2. * THIRTY_RAD + EPSILON_RAD) {
        return nu_60(ra, a_gi);
    } else if (gamma < 3. * THIRTY_RAD - EPSILON_RAD) {
        return nu_60_90(ra, gamma, a_gi, err);
    } else if (gamma < 3. * THIRTY_RAD + EPSILON_RAD) {
        return nu_90(ra, a_gi, err);
    } else if (gamma < 6. * THIRTY_RAD) {
        return nu_90_180(ra, a_gi, gamma, err);
    }
    if (err) *err = OR_ERR_UNREACHABLE;
    return NAN;
}

/* gas.rs:126-152 */
double or_cavity_convection(int gas, double height, double thickness, double gamma,
                            double t_front, double t_back, int *err) {
    if (t_front > t_back) gamma = or_to_radians(180.) - gamma;
    double a_gi = height / thickness;
    double ra = or_raleigh(gas, t_front, t_back, thickness);
    double nu = or_nusselt(ra, gamma, a_gi, err);
    double temp = (in_kelvin(t_front) + in_kelvin(t_back)) / 2.;
    double lambda = or_gas_thermal_conductivity(gas, temp);
    return nu * lambda / thickness;
}

/* cavity.rs:59-69 */
double or_cavity_u_value(const or_cavity *c, double t_front, double t_back, int *err) {
    double conv = or_cavity_convection(c->gas, c->height, c->thickness, c->angle, t_front, t_back, err);
    double tm = (t_back + t_front) / 2. + 273.15;
    double rad = 4. * or_powi(tm, 3) * OR_SIGMA * c->ein * c->eout / (1. - (1. - c->ein) * (1. - c->eout));
    return rad + conv;
}

/* zone.rs:59-65 */
double or_zone_mcp(double volume, double temp) {
    double air_density = or_gas_density(OR_AIR, temp + 273.15);
    double air_specific_heat = or_gas_heat_capacity(OR_AIR, temp + 273.15);
    return volume * air_density * air_specific_heat / 1.;
}

/* ------------------------------------------------------------------ */
/* matrix::Matrix::prod_tri_diag_into — call sites surface.rs:268,280,286,292 */
void or_prod_tri_diag(int n, const double *lo, const double *dg, const double *up,
                      const double *x, double *y) {
    for (int i = 0; i < n; i++) {
        double acc = 0.0;
        if (i > 0) acc += lo[i] * x[i - 1];
        acc += dg[i] * x[i];
        if (i < n - 1) acc += up[i] * x[i + 1];
        y[i] = acc;
    }
}

/* matrix::Matrix::mut_n_diag_gaussian(rhs, 3) — call site surface.rs:834.
 * Banded Gaussian elimination without pivoting followed by back-substitution. */
void or_tri_diag_gaussian(int n, double *lo, double *dg, double *up, double *rhs, double *x) {
    for (int i = 1; i < n; i++) {
        double f = lo[i] / dg[i - 1];
        dg[i] -= f * up[i - 1];
        rhs[i] -= f * rhs[i - 1];
    }
    x[n - 1] = rhs[n - 1] / dg[n - 1];
    for (int i = n - 2; i >= 0; i--) {
        x[i] = (rhs[i] - up[i] * x[i + 1]) / dg[i];
    }
}

/* surface.rs:168-187 */
void or_rearrange_k(int n, double dt, const double *c, double *lo, double *dg, double *up, double *q) {
    for (int r = 0; r < n; r++) {
        double v = dt / c[r];
        if (r > 0) lo[r] *= v;
        dg[r] *= v;
        if (r < n - 1) up[r] *= v;
        q[r] *= v;
    }
}

/* surface.rs:228-308 */
void or_rk4(int n, const double *lo, const double *dg, const double *up, const double *q, double *temps) {
    double k1[OR_MAX_NODES], k2[OR_MAX_NODES], k3[OR_MAX_NODES], k4[OR_MAX_NODES], aux[OR_MAX_NODES];
    int i;
    /* :268-269 */
    or_prod_tri_diag(n, lo, dg, up, temps, k1);
    for (i = 0; i < n; i++) k1[i] += q[i];
    /* :276-277 */
    for (i = 0; i < n; i++) aux[i] = k1[i] * 0.5;
    for (i = 0; i < n; i++) aux[i] += temps[i];
    /* :280-281 */
    or_prod_tri_diag(n, lo, dg, up, aux, k2);
    for (i = 0; i < n; i++) k2[i] += q[i];
    /* :284-287 */
    for (i = 0; i < n; i++) aux[i] = k2[i] * 0.5;
    for (i = 0; i < n; i++) aux[i] += temps[i];
    or_prod_tri_diag(n, lo, dg, up, aux, k3);
    for (i = 0; i < n; i++) k3[i] += q[i];
    /* :290-293 */
    for (i = 0; i < n; i++) aux[i] = k3[i];
    for (i = 0; i < n; i++) aux[i] += temps[i];
    or_prod_tri_diag(n, lo, dg, up, aux, k4);
    for (i = 0; i < n; i++) k4[i] += q[i];
    /* :296-299 */
    for (i = 0; i < n; i++) k1[i] /= 6.;
    for (i = 0; i < n; i++) k2[i] /= 3.;
    for (i = 0; i < n; i++) k3[i] /= 3.;
    for (i = 0; i < n; i++) k4[i] /= 6.;
    /* :302-305 */
    for (i = 0; i < n; i++) temps[i] += k1[i];
    for (i = 0; i < n; i++) temps[i] += k2[i];
    for (i = 0; i < n; i++) temps[i] += k3[i];
    for (i = 0; i < n; i++) temps[i] += k4[i];
}

/* discretization.rs:48-55 */
static double seg_u_value(const double *uvalue, const int32_t *seg_cavity, const or_cavity *cavities,
                          int g, double t_before, double t_after, int *err) {
    if (seg_cavity && seg_cavity[g] >= 0) return or_cavity_u_value(&cavities[seg_cavity[g]], t_before, t_after, err);
    return uvalue[g]; /* Solid(u) -> u ; Back -> 0 */
}

/* discretization.rs:596-700 */
int or_get_k_q(int nrows, const double *uvalue, const int32_t *seg_cavity, const or_cavity *cavities,
               int ini, int fin, const double *temperatures,
               double front_air_t, double front_rad_t, double front_hs, double front_rad_hs,
               double back_air_t, double back_rad_t, double back_hs, double back_rad_hs,
               double *lo, double *dg, double *up, double *q) {
    int err = 0;
    int nnodes = fin - ini;
    /* :622-623 */
    for (int i = 0; i < nnodes; i++) { lo[i] = 0.0; dg[i] = 0.0; up[i] = 0.0; q[i] = 0.0; }
    /* :634-655 */
    for (int local_i = 0; local_i < nnodes - 1; local_i++) {
        int global_i = ini + local_i;
        double t_this = temperatures[global_i];
        double t_next = (global_i + 1 < nrows) ? temperatures[global_i + 1] : back_air_t; /* get_t_after :627-632 */
        double u = seg_u_value(uvalue, seg_cavity, cavities, global_i, t_this, t_next, &err);
        dg[local_i] += -u;
        dg[local_i + 1] = dg[local_i + 1] - u;
        up[local_i] = up[local_i] + u;
        lo[local_i + 1] = lo[local_i + 1] + u;
    }
    /* :658-677 */
    double hs_front, front_q;
    if (ini == 0) {
        double ts = temperatures[0];
        front_q = front_air_t * front_hs + front_rad_hs * (front_rad_t - ts);
        hs_front = front_hs;
    } else {
        double t_before = temperatures[ini - 1];
        double t_after = temperatures[ini];
        double u = seg_u_value(uvalue, seg_cavity, cavities, ini - 1, t_before, t_after, &err);
        hs_front = u;
        front_q = u * t_before;
    }
    q[0] += front_q;
    dg[0] += -hs_front;
    /* :680-697 */
    double hs_back, back_q;
    if (fin == nrows) {
        double ts = temperatures[fin - 1];
        back_q = back_air_t * back_hs + back_rad_hs * (back_rad_t - ts);
        hs_back = back_hs;
    } else {
        double t_before = temperatures[fin - 1];
        double t_after = (fin < nrows) ? temperatures[fin] : back_air_t;
        double u = seg_u_value(uvalue, seg_cavity, cavities, fin - 1, t_before, t_after, &err);
        hs_back = u;
        back_q = u * t_after;
    }
    q[nnodes - 1] += back_q;
    dg[nnodes - 1] += -hs_back;
    return err;
}

/* discretization.rs:117-160 */
static int chunk_segments(const int *indexes, int n_idx, int *out) {
    if (n_idx == 0) return 0;
    int start = indexes[0], prev = start, n_out = 0;
    for (int j = 1; j < n_idx; j++) {
        int i = indexes[j];
        if (i - prev == 1) {
            prev = i;
        } else {
            out[2 * n_out] = start; out[2 * n_out + 1] = prev + 1; n_out++;
            start = i; prev = start;
        }
    }
    out[2 * n_out] = start; out[2 * n_out + 1] = prev + 1; n_out++;
    return n_out;
}

void or_get_chunks(int nrows, const double *mass, int *n_mass, int *mass_chunks,
                   int *n_nomass, int *nomass_chunks) {
    int mass_nodes[OR_MAX_NODES], nomass_nodes[OR_MAX_NODES];
    int nm = 0, nn = 0;
    for (int i = 0; i < nrows; i++) if (mass[i] >= 1e-5) mass_nodes[nm++] = i;  /* :149 */
    for (int i = 0; i < nrows; i++) if (mass[i] < 1e-5) nomass_nodes[nn++] = i; /* :155 */
    *n_mass = chunk_segments(mass_nodes, nm, mass_chunks);
    *n_nomass = chunk_segments(nomass_nodes, nn, nomass_chunks);
}

/* ------------------------------------------------------------------ */
typedef struct {
    double air_t, air_speed, rad_t, surf_t, cos_tilt;
} or_env; /* convection.rs:27-52 (roughness_index is the constant 1) */

/* surface.rs:596-717 */
static int calc_border_conditions(const or_model *m, int64_t s, const double *state,
                                  double t_front, double t_back, double wind_direction, double wind_speed,
                                  or_env *front_env, or_env *back_env, double *front_hs, double *back_hs) {
    int err = 0;
    int64_t first = m->first_node_slot[s];
    int64_t last = first + (m->node_offset[s + 1] - m->node_offset[s]) - 1;
    double ir_front = state[m->ir_front_slot[s]];
    double ir_back = state[m->ir_back_slot[s]];
    double cos_tilt = m->cos_tilt[s];
    int windward = or_is_windward(wind_direction, cos_tilt, m->normal_x[s], m->normal_y[s]);
    double fhs, bhs;

    switch (m->front_kind[s]) {
    case OR_SPACE: /* :612-626 */
        front_env->air_t = t_front; front_env->air_speed = 0.0; front_env->rad_t = t_front;
        front_env->surf_t = state[first]; front_env->cos_tilt = cos_tilt;
        fhs = or_tarp_natural(front_env->air_t, front_env->surf_t, front_env->cos_tilt, &err);
        break;
    case OR_AMBIENT: /* :627-641 */
        front_env->air_t = m->front_ambient[s]; front_env->air_speed = 0.0; front_env->rad_t = t_front;
        front_env->surf_t = state[first]; front_env->cos_tilt = cos_tilt;
        fhs = or_tarp_natural(front_env->air_t, front_env->surf_t, front_env->cos_tilt, &err);
        break;
    case OR_OUTDOOR: /* :643-657 */
        front_env->air_t = t_front; front_env->air_speed = wind_speed * m->wind_modifier[s];
        front_env->rad_t = pow(ir_front / OR_SIGMA, 0.25) - 273.15;
        front_env->surf_t = state[first]; front_env->cos_tilt = -cos_tilt; /* :652 */
        fhs = or_tarp_total(front_env->air_t, front_env->surf_t, front_env->cos_tilt, front_env->air_speed,
                            m->area[s], m->perimeter[s], windward, &err);
        break;
    default: return OR_ERR_GROUND;
    }

    switch (m->back_kind[s]) {
    case OR_SPACE: /* :661-671 */
        back_env->air_t = t_back; back_env->air_speed = 0.0; back_env->rad_t = t_back;
        back_env->surf_t = state[last]; back_env->cos_tilt = cos_tilt;
        bhs = or_tarp_natural(back_env->air_t, back_env->surf_t, back_env->cos_tilt, &err);
        break;
    case OR_AMBIENT: /* :672-686 — uses t_front and the FRONT temperature (reproduced as is) */
        back_env->air_t = m->back_ambient[s]; back_env->air_speed = 0.0; back_env->rad_t = t_front;
        back_env->surf_t = state[first]; back_env->cos_tilt = cos_tilt;
        bhs = or_tarp_natural(back_env->air_t, back_env->surf_t, back_env->cos_tilt, &err);
        break;
    case OR_OUTDOOR: /* :688-701 */
        back_env->air_t = t_back; back_env->air_speed = wind_speed * m->wind_modifier[s];
        back_env->rad_t = pow(ir_back / OR_SIGMA, 0.25) - 273.15;
        back_env->surf_t = state[last]; back_env->cos_tilt = cos_tilt;
        bhs = or_tarp_total(back_env->air_t, back_env->surf_t, back_env->cos_tilt, back_env->air_speed,
                            m->area[s], m->perimeter[s], windward, &err);
        break;
    default: return OR_ERR_GROUND;
    }
    if (err) return err;
    if (isnan(fhs) || isnan(bhs)) return OR_ERR_NAN_HS; /* :704-707 */
    /* :708-714 debug overrides */
    if (m->front_hs_fix && !isnan(m->front_hs_fix[s])) fhs = m->front_hs_fix[s];
    if (m->back_hs_fix && !isnan(m->back_hs_fix[s])) bhs = m->back_hs_fix[s];
    *front_hs = fhs;
    *back_hs = bhs;
    return 0;
}

typedef struct {
    const or_model *m;
    int64_t s;
    int nrows;
    const double *mass, *uvalue;
    const int32_t *seg_cavity;
    const double *state;
    double t_front, t_back, wind_direction, wind_speed;
} or_ctx;

/* surface.rs:720-787 */
static int march_mass(const or_ctx *c, double *temps, const double *solar, double dt,
                      double front_rad_hs, double back_rad_hs, int ini, int fin) {
    double lo[OR_MAX_NODES], dg[OR_MAX_NODES], up[OR_MAX_NODES], q[OR_MAX_NODES], cc[OR_MAX_NODES], local[OR_MAX_NODES];
    or_env fe, be;
    double fhs, bhs;
    int n = fin - ini;
    int err = calc_border_conditions(c->m, c->s, c->state, c->t_front, c->t_back, c->wind_direction, c->wind_speed, &fe, &be, &fhs, &bhs);
    if (err) return err;
    err = or_get_k_q(c->nrows, c->uvalue, c->seg_cavity, c->m->cavities, ini, fin, temps,
                     fe.air_t, fe.rad_t, fhs, front_rad_hs, be.air_t, be.rad_t, bhs, back_rad_hs, lo, dg, up, q);
    if (err) return err;
    for (int i = 0; i < n; i++) cc[i] = c->mass[ini + i];          /* :753-763 */
    for (int i = 0; i < n; i++) q[i] += solar[ini + i];             /* :766-769 */
    or_rearrange_k(n, dt, cc, lo, dg, up, q);                       /* :771 */
    for (int i = 0; i < n; i++) local[i] = temps[ini + i];          /* :775-778 */
    or_rk4(n, lo, dg, up, q, local);                                /* :780 */
    for (int i = 0; i < n; i++) temps[ini + i] = local[i];          /* :782-785 */
    return 0;
}

/* surface.rs:790-898 */
static int march_nomass(const or_ctx *c, double *temps, const double *solar,
                        double front_rad_hs, double back_rad_hs, int ini, int fin, int64_t *iters) {
    double lo[OR_MAX_NODES], dg[OR_MAX_NODES], up[OR_MAX_NODES], q[OR_MAX_NODES], x[OR_MAX_NODES];
    double old_err = 99999.;
    long count = 0;
    int n = fin - ini;
    for (;;) {
        or_env fe, be;
        double fhs, bhs;
        int err = calc_border_conditions(c->m, c->s, c->state, c->t_front, c->t_back, c->wind_direction, c->wind_speed, &fe, &be, &fhs, &bhs);
        if (err) return err;
        err = or_get_k_q(c->nrows, c->uvalue, c->seg_cavity, c->m->cavities, ini, fin, temps,
                         fe.air_t, fe.rad_t, fhs, front_rad_hs, be.air_t, be.rad_t, bhs, back_rad_hs, lo, dg, up, q);
        if (err) return err;
        if (iters) (*iters)++;
        for (int i = 0; i < n; i++) q[i] += solar[ini + i]; /* :828-831 */
        for (int i = 0; i < n; i++) q[i] *= -1.;             /* :832 */
        or_tri_diag_gaussian(n, lo, dg, up, q, x);           /* :834 */
        double e = 0.0;
        for (int i = 0; i < n; i++) e += fabs(x[i] - temps[ini + i]); /* :836-841 */
        if (e > old_err) break;                              /* :842-848 */
        if (isnan(e)) return OR_ERR_NAN_NOMASS;              /* :850 */
        for (int i = 0; i < n; i++) {                        /* :878-883 */
            temps[ini + i] += x[i];
            temps[ini + i] *= 0.5;
        }
        double max_allowed_error = (count < 100) ? 0.01 : 0.5; /* :885 */
        if (e / (double)n < max_allowed_error) break;        /* :887-893 */
        old_err = e;
        count += 1;
    }
    return 0;
}

/* surface.rs:902-1024 */
static int surface_march(const or_ctx *c, double dt, double *temps, int64_t *iters) {
    const or_model *m = c->m;
    int64_t s = c->s;
    int nrows = c->nrows;
    int64_t off = m->node_offset[s];
    double solar[OR_MAX_NODES];
    int mass_chunks[2 * OR_MAX_NODES], nomass_chunks[2 * OR_MAX_NODES], n_mass, n_nomass;
    int err;

    for (int i = 0; i < nrows; i++) temps[i] = c->state[m->first_node_slot[s] + i]; /* :912-913 */

    /* :916-923 */
    double solar_front = c->state[m->solar_front_slot[s]];
    if (isnan(solar_front) || solar_front < 0.0) solar_front = 0.0;
    double solar_back = c->state[m->solar_back_slot[s]];
    if (isnan(solar_back) || solar_front < 0.0) solar_back = 0.0; /* sic: tests solar_front */

    /* :930-931 */
    for (int i = 0; i < nrows; i++) solar[i] = m->front_alpha[off + i] * solar_front;
    for (int i = 0; i < nrows; i++) solar[i] += m->back_alpha[off + i] * solar_back;

    or_get_chunks(nrows, c->mass, &n_mass, mass_chunks, &n_nomass, nomass_chunks);

    /* :939-948 */
    or_env fe, be;
    double fhs, bhs;
    err = calc_border_conditions(m, s, c->state, c->t_front, c->t_back, c->wind_direction, c->wind_speed, &fe, &be, &fhs, &bhs);
    if (err) return err;
    double front_rad_hs = 4. * m->front_emissivity[s] * OR_SIGMA * or_powi(273.15 + (fe.rad_t + fe.surf_t) / 2., 3);
    double back_rad_hs = 4. * m->back_emissivity[s] * OR_SIGMA * or_powi(273.15 + (be.rad_t + be.surf_t) / 2., 3);

    /* :950-965 */
    for (int ci = 0; ci < n_nomass; ci++) {
        err = march_nomass(c, temps, solar, front_rad_hs, back_rad_hs, nomass_chunks[2 * ci], nomass_chunks[2 * ci + 1], iters);
        if (err) return err;
    }

    /* :969-978 (same values again: `state` has not changed) */
    err = calc_border_conditions(m, s, c->state, c->t_front, c->t_back, c->wind_direction, c->wind_speed, &fe, &be, &fhs, &bhs);
    if (err) return err;
    front_rad_hs = 4. * m->front_emissivity[s] * OR_SIGMA * or_powi(273.15 + (fe.rad_t + fe.surf_t) / 2., 3);
    back_rad_hs = 4. * m->back_emissivity[s] * OR_SIGMA * or_powi(273.15 + (be.rad_t + be.surf_t) / 2., 3);

    /* :984-1000 */
    for (int ci = 0; ci < n_mass; ci++) {
        err = march_mass(c, temps, solar, dt, front_rad_hs, back_rad_hs, mass_chunks[2 * ci], mass_chunks[2 * ci + 1]);
        if (err) return err;
    }
    return 0;
}

/* model.rs:79-96 */
static int boundary_temperature(const or_model *m, int kind, int zone, double ambient, double t_out,
                                const double *state, double *out) {
    switch (kind) {
    case OR_SPACE: *out = state[m->zone_slot[zone]]; return 0;
    case OR_AMBIENT: *out = ambient; return 0;
    case OR_OUTDOOR: *out = t_out; return 0;
    default: return OR_ERR_GROUND;
    }
}

/* model.rs:120-171: the body of iterate_surfaces for one surface */
static int iterate_one(const or_model *m, double *state, int64_t s, double wind_direction, double wind_speed,
                       double t_out, int64_t *iters) {
    double temps[OR_MAX_NODES];
    double t_front, t_back;
    int err;
    int nrows = (int)(m->node_offset[s + 1] - m->node_offset[s]);
    if (nrows > OR_MAX_NODES) return -100;
    err = boundary_temperature(m, m->front_kind[s], m->front_zone[s], m->front_ambient[s], t_out, state, &t_front); /* :124-125 */
    if (err) return err;
    err = boundary_temperature(m, m->back_kind[s], m->back_zone[s], m->back_ambient[s], t_out, state, &t_back); /* :126-127 */
    if (err) return err;

    or_ctx c;
    c.m = m; c.s = s; c.nrows = nrows;
    c.mass = m->mass + m->node_offset[s];
    c.uvalue = m->uvalue + m->node_offset[s];
    c.seg_cavity = m->seg_cavity ? m->seg_cavity + m->node_offset[s] : NULL;
    c.state = state;
    c.t_front = t_front; c.t_back = t_back; c.wind_direction = wind_direction; c.wind_speed = wind_speed;

    err = surface_march(&c, m->dt, temps, iters); /* :130-138 */
    if (err) return err;

    for (int i = 0; i < nrows; i++) state[m->first_node_slot[s] + i] = temps[i]; /* :145-147 */

    double ts_front = temps[0], ts_back = temps[nrows - 1]; /* :150-151 */
    or_env fe, be;
    double fhs, bhs;
    err = calc_border_conditions(m, s, state, t_front, t_back, wind_direction, wind_speed, &fe, &be, &fhs, &bhs); /* :152-153 */
    if (err) return err;
    state[m->hs_front_slot[s]] = fhs; /* :154-159 */
    state[m->hs_back_slot[s]] = bhs;
    state[m->flow_front_slot[s]] = (ts_front - t_front) * fhs; /* :161,164-166 */
    state[m->flow_back_slot[s]] = (ts_back - t_back) * bhs;    /* :162,167-169 */
    return 0;
}

/* model.rs:102-180 */
int or_iterate_surfaces(const or_model *m, double *state, int64_t s0, int64_t s1,
                        double wind_direction, double wind_speed, double t_out, int64_t *iters) {
    for (int64_t s = s0; s < s1; s++) {
        int err = iterate_one(m, state, s, wind_direction, wind_speed, t_out, iters);
        if (err) return err;
    }
    return 0;
}

/* model.rs:489-597 (surface part :556-590; capacitance :549-552). a, b arrive holding the host terms. */
void or_zones_abc(const or_model *m, const double *state, double *a, double *b, double *c) {
    for (int64_t z = 0; z < m->n_zones; z++) c[z] = or_zone_mcp(m->zone_volume[z], state[m->zone_slot[z]]);
    for (int64_t s = 0; s < m->n_surfaces; s++) {
        double h_front = state[m->hs_front_slot[s]];
        double h_back = state[m->hs_back_slot[s]];
        double ai = m->area[s];
        int64_t first = m->first_node_slot[s];
        int64_t last = first + (m->node_offset[s + 1] - m->node_offset[s]) - 1;
        if (m->front_kind[s] == OR_SPACE) {
            int z = m->front_zone[s];
            double temp = state[first];
            a[z] += h_front * ai * temp;
            b[z] += h_front * ai;
        }
        if (m->back_kind[s] == OR_SPACE) {
            int z = m->back_zone[s];
            double temp = state[last];
            a[z] += h_back * ai * temp;
            b[z] += h_back * ai;
        }
    }
}

/* model.rs:410-423 + 650-674 */
static int zones_update(const or_model *m, double *state, const double *t_current,
                        const double *zone_a0, const double *zone_b0,
                        double *a, double *b, double *c) {
    for (int64_t z = 0; z < m->n_zones; z++) {
        a[z] = zone_a0 ? zone_a0[z] : 0.0;
        b[z] = zone_b0 ? zone_b0[z] : 0.0;
    }
    or_zones_abc(m, state, a, b, c);
    for (int64_t z = 0; z < m->n_zones; z++) {
        double ft;
        if (fabs(b[z]) > 1e-9) {
            ft = a[z] / b[z] + (t_current[z] - a[z] / b[z]) * exp(-b[z] * m->dt / c[z]);
        } else {
            ft = t_current[z];
        }
        if (isnan(ft)) return OR_ERR_NAN_ZONE;
        state[m->zone_slot[z]] = ft;
    }
    return 0;
}

/* model.rs:359-427 */
int or_model_march(const or_model *m, double *state, const double *weather, int n_sub,
                   const double *zone_a0, const double *zone_b0, int64_t *nomass_iterations) {
    int err = 0;
    int64_t nz = m->n_zones > 0 ? m->n_zones : 1;
    double *buf = (double *)malloc(sizeof(double) * 4 * (size_t)nz);
    double *t_current = buf, *a = buf + nz, *b = buf + 2 * nz, *c = buf + 3 * nz;
    if (nomass_iterations) *nomass_iterations = 0;
    for (int step = 0; step < n_sub && !err; step++) {
        double t_out = weather[3 * step + 0];
        double wind_direction = weather[3 * step + 1];
        double wind_speed = weather[3 * step + 2];
        for (int64_t z = 0; z < m->n_zones; z++) t_current[z] = state[m->zone_slot[z]]; /* :385 */
        err = or_iterate_surfaces(m, state, 0, m->n_surfaces, wind_direction, wind_speed, t_out, nomass_iterations); /* :388-408 */
        if (err) break;
        err = zones_update(m, state, t_current, zone_a0, zone_b0, a, b, c); /* :412-423 */
    }
    free(buf);
    return err;
}

int or_model_march_mt(const or_model *m, double *state, const double *weather, int n_sub,
                      const double *zone_a0, const double *zone_b0, int n_threads) {
    int err = 0;
    int64_t nz = m->n_zones > 0 ? m->n_zones : 1;
    double *buf = (double *)malloc(sizeof(double) * 4 * (size_t)nz);
    double *t_current = buf, *a = buf + nz, *b = buf + 2 * nz, *c = buf + 3 * nz;
    (void)n_threads;
    for (int step = 0; step < n_sub && !err; step++) {
        double t_out = weather[3 * step + 0];
        double wind_direction = weather[3 * step + 1];
        double wind_speed = weather[3 * step + 2];
        for (int64_t z = 0; z < m->n_zones; z++) t_current[z] = state[m->zone_slot[z]];
        int first_err = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
        for (int64_t s = 0; s < m->n_surfaces; s++) {
            int e = iterate_one(m, state, s, wind_direction, wind_speed, t_out, NULL);
            if (e != 0) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                first_err = e;
            }
        }
        if (first_err) { err = first_err; break; }
        err = zones_update(m, state, t_current, zone_a0, zone_b0, a, b, c);
    }
    free(buf);
    return err;
}

/* ------------------------------------------------------------------ */
/* discretization.rs:410-544 */
int or_discretize_construction(int n_layers, const or_layer *layers, double model_dt,
                               double max_dx, double min_dt, int *n_elements) {
    const double MAX_RS = 0.05; /* discretization.rs:21 */
    int n = 1;
restart:;
    double dt = model_dt / (double)n;
    for (int l = 0; l < n_layers; l++) {
        const or_layer *L = &layers[l];
        if (L->is_gas) { n_elements[l] = 0; continue; } /* :447-450 */
        double thickness = L->thickness, k = L->k, rho = L->rho, cp = L->cp;
        double a_coef = 1.;
        double b_coef = -dt / (rho * cp * MAX_RS);
        double c_coef = -2. * dt * k / (rho * cp);
        double disc = b_coef * b_coef - 4. * a_coef * c_coef;
        double min_dx = (-b_coef + sqrt(disc)) / (2. * a_coef);
        if (min_dx > thickness) {
            double next_dt = model_dt / (double)(n + 1);
            if (next_dt > min_dt) { n += 1; goto restart; }
            n_elements[l] = 0;
        } else {
            double mm = floor(thickness / min_dx);
            double dx = thickness / mm;
            if (dx > max_dx) {
                double next_dt = model_dt / (double)(n + 1);
                if (next_dt > min_dt) { n += 1; goto restart; }
                n_elements[l] = 0;
            } else {
                n_elements[l] = (int)mm;
            }
        }
    }
    return n;
}

/* discretization.rs:174-176 */
int or_count_nodes(int n_layers, const int *n_elements) {
    int n_nodes = 0, zeros = 0;
    for (int l = 0; l < n_layers; l++) { n_nodes += n_elements[l]; if (n_elements[l] == 0) zeros++; }
    return n_nodes + zeros + 1;
}

/* discretization.rs:163-298 */
int or_build(int n_layers, const or_layer *layers, const int *n_elements, double height, double angle,
             double *mass_out, double *uvalue, int32_t *seg_cavity, or_cavity *cavities_out, int cav_base) {
    int n_nodes = or_count_nodes(n_layers, n_elements);
    int n_cav = 0;
    for (int i = 0; i < n_nodes; i++) { mass_out[i] = 0.0; uvalue[i] = NAN; seg_cavity[i] = -1; } /* UValue::None */
    int n_segment = 0;
    for (int l = 0; l < n_layers; l++) {
        int n = n_elements[l];
        const or_layer *L = &layers[l];
        double mass;
        if (n == 0) {
            mass = 0.0;
        } else if (!L->is_gas) {
            double dx = L->thickness / (double)n;
            mass = L->rho * L->cp * dx;
        } else {
            mass = 0.0;
        }
        if (n == 0) n = 1;
        for (int e = 0; e < n; e++) {
            if (!L->is_gas) {
                mass_out[n_segment] += mass / 2.;
                mass_out[n_segment + 1] += mass / 2.;
                double dx = L->thickness / (double)n;
                uvalue[n_segment] = L->k / dx;
            } else {
                if (l == 0) return -1;            /* gas as first layer :242-248 */
                if (l + 1 >= n_layers) return -2; /* gas as last layer :252-260 */
                if (layers[l + 1].is_gas || layers[l - 1].is_gas) return -3; /* :266-274 */
                or_cavity *c = &cavities_out[n_cav];
                c->gas = L->gas;
                c->thickness = L->thickness;
                c->height = height;
                c->angle = angle;
                c->eout = layers[l - 1].back_thermal_abs;
                c->ein = layers[l + 1].front_thermal_abs;
                c->pad_ = 0;
                uvalue[n_segment] = 0.0;
                seg_cavity[n_segment] = cav_base + n_cav;
                n_cav++;
            }
            n_segment++;
        }
        uvalue[n_nodes - 1] = 0.0; /* UValue::Back :290 */
        seg_cavity[n_nodes - 1] = -1;
    }
    return n_cav;
}

/* ---- glazing.rs ---- */
typedef struct { double tau, rho_front, rho_back, alpha_front, alpha_back; } or_glazing;

static or_glazing glazing_new(double tau, double rho_front, double rho_back) { /* glazing.rs:50-65 */
    or_glazing g;
    g.tau = tau; g.rho_front = rho_front; g.rho_back = rho_back;
    g.alpha_front = 1. - tau - rho_front;
    g.alpha_back = 1. - tau - rho_back;
    return g;
}
static or_glazing glazing_combine(const or_glazing *s, const or_glazing *o) { /* glazing.rs:180-221 */
    double rho_back = o->rho_back + or_powi(o->tau, 2) * s->rho_back / (1. - o->rho_front * s->rho_back);
    double rho_front = s->rho_front + or_powi(s->tau, 2) * o->rho_front / (1. - s->rho_back * o->rho_front);
    double tau = s->tau * o->tau / (1. - s->rho_back * o->rho_front);
    return glazing_new(tau, rho_front, rho_back);
}
static or_glazing glazing_combine_layers(const or_glazing *layers, int n) { /* glazing.rs:223-233 */
    if (n == 1) return layers[0];
    or_glazing rest = glazing_combine_layers(layers + 1, n - 1);
    return glazing_combine(&layers[0], &rest);
}
static void glazing_combined_alphas(const or_glazing *s, const or_glazing *o, double *a1, double *a2) { /* glazing.rs:247-252 */
    double denom = 1. - s->rho_back * o->rho_front;
    *a1 = s->alpha_front + s->alpha_back * s->tau * o->rho_front / denom;
    *a2 = o->alpha_front * s->tau / denom;
}
static int glazing_alphas(const or_glazing *layers, int n, double *ret) { /* glazing.rs:259-286 */
    if (n == 0) return 0;
    if (n == 1) { ret[0] = layers[0].alpha_front; return 1; }
    double acc_alpha = 0.0;
    int k = 0;
    for (int i = 1; i < n; i++) {
        or_glazing g0 = glazing_combine_layers(layers, i);
        or_glazing g1 = glazing_combine_layers(layers + i, n - i);
        double a0, unused;
        glazing_combined_alphas(&g0, &g1, &a0, &unused);
        ret[k++] = a0 - acc_alpha;
        acc_alpha = a0;
    }
    or_glazing g0 = glazing_combine_layers(layers, n - 1);
    double unused, a1;
    glazing_combined_alphas(&g0, &layers[n - 1], &unused, &a1);
    ret[k++] = a1;
    return k;
}

int or_glazing_alphas(int n, const double *tau, const double *rho_front, const double *rho_back, double *alphas) {
    or_glazing g[64];
    if (n > 64) return -1;
    for (int i = 0; i < n; i++) g[i] = glazing_new(tau[i], rho_front[i], rho_back[i]);
    return glazing_alphas(g, n, alphas);
}

void or_glazing_combine_layers(int n, const double *tau, const double *rho_front, const double *rho_back, double *out5) {
    or_glazing g[64];
    for (int i = 0; i < n && i < 64; i++) g[i] = glazing_new(tau[i], rho_front[i], rho_back[i]);
    or_glazing r = glazing_combine_layers(g, n);
    out5[0] = r.tau; out5[1] = r.rho_front; out5[2] = r.rho_back; out5[3] = r.alpha_front; out5[4] = r.alpha_back;
}

/* glazing.rs:67-112 with `order` = layer indices front->back or back->front */
static int glazing_system(int n_layers, const or_layer *layers, int reverse, or_glazing *out) {
    int n = 0, pos = 0;
    for (;;) {
        if (pos >= n_layers) return -10; /* i.next().unwrap() on an exhausted iterator */
        const or_layer *L = &layers[reverse ? n_layers - 1 - pos : pos];
        pos++;
        if (L->is_gas) return -11; /* "NOT expecting a gas" */
        double rho_front = 1. - L->tau - L->front_solar_abs;
        double rho_back = 1. - L->tau - L->back_solar_abs;
        out[n++] = glazing_new(L->tau, rho_front, rho_back);
        if (L->tau < 1e-9) break;
        if (pos < n_layers) {
            const or_layer *G = &layers[reverse ? n_layers - 1 - pos : pos];
            pos++;
            if (!G->is_gas) return -12; /* "Expecting a Gas" */
        } else {
            break;
        }
    }
    return n;
}

/* surface.rs:463-537 */
int or_node_alphas(int n_layers, const or_layer *layers, const int *n_elements, int n_nodes,
                   double *front_alphas, double *back_alphas) {
    or_glazing sys[64];
    double prev[64];
    if (n_layers > 64) return -1;
    int ng = glazing_system(n_layers, layers, 0, sys);
    if (ng < 0) return ng;
    int na = glazing_alphas(sys, ng, prev);
    if (na != 1 && na != n_layers) return -20; /* mixture panic :470-472 */
    for (int i = 0; i < n_nodes; i++) { front_alphas[i] = 0.0; back_alphas[i] = 0.0; }
    int global_i = 0;
    for (int ai = 0; ai < na; ai++) {
        int layer_index = 2 * ai;
        int n = n_elements[layer_index] == 0 ? 1 : n_elements[layer_index];
        double tau = layers[layer_index].tau;
        if (layers[layer_index].is_gas) return -21; /* unreachable!() */
        if (tau > 0.0) {
            for (int li = 0; li <= n; li++) front_alphas[global_i + li] += prev[ai] / (double)(n + 1);
        } else {
            front_alphas[global_i] += prev[ai];
        }
        global_i += n + 1;
    }
    ng = glazing_system(n_layers, layers, 1, sys);
    if (ng < 0) return ng;
    na = glazing_alphas(sys, ng, prev);
    if (na != 1 && na != n_layers) return -20; /* :506-508 */
    global_i = n_nodes;
    for (int ai = 0; ai < na; ai++) {
        int layer_index = n_layers - 2 * ai - 1;
        int n = n_elements[layer_index] == 0 ? 1 : n_elements[layer_index];
        double tau = layers[layer_index].tau;
        if (layers[layer_index].is_gas) return -21;
        if (tau > 0.0) {
            for (int li = 0; li <= n; li++) back_alphas[global_i - li - 1] += prev[ai] / (double)(n + 1);
        } else {
            back_alphas[global_i - 1] += prev[ai];
        }
        global_i -= n + 1;
    }
    return 0;
}
