// device_math.hpp — leaf physics of the wall heat-conduction path as gfx950 device functions.
// Each function cites the reference lines (relative to the reference repo root) it implements.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.hpp"

namespace heat {

constexpr double kPi = 3.14159265358979323846264338327950288;

// TARP natural convection — reference src/convection.rs:87-110.
// |dT|^(1/3) is evaluated with cbrt(): relative difference to powf(1./3.) is
// |ln x| * 1.9e-17, far below the 1e-9 parity tolerance.
__device__ __forceinline__ double tarp_natural(double air_t, double surf_t, double cos_tilt, int &bad) {
    const double delta_t = air_t - surf_t;
    const double adt = fabs(delta_t);
    const double act = fabs(cos_tilt);
    const double c3 = cbrt(adt);
    double h;
    if (adt < 1e-3 || act < 1e-3) {
        h = 1.31 * c3;
    } else if ((delta_t < 0. && cos_tilt < 0.) || (delta_t > 0. && cos_tilt > 0.)) {
        h = 9.482 * c3 / (7.238 - act);
    } else if ((delta_t > 0. && cos_tilt < 0.) || (delta_t < 0. && cos_tilt > 0.)) {
        h = 1.81 * c3 / (1.382 + act);
    } else {  // unreachable!() in the reference (NaN temperatures)
        bad |= FLAG_UNREACHABLE;
        h = __builtin_nan("");
    }
    return (h < 0.1) ? 0.1 : h;  // MIN_H, convection.rs:22,105-109
}

// The same with the tilt-dependent factors taken out (fast classes): nat_pos / nat_neg = h / |dT|^(1/3) for air
// warmer / colder than the surface, set up on the host from the branches above. c3 * (9.482 / (7.238 - |cos|))
// instead of (9.482 * c3) / (7.238 - |cos|): one rounding more, two divisions less per evaluation.
__device__ __forceinline__ double tarp_natural_coef(double air_t, double surf_t, double nat_pos, double nat_neg, int &bad) {
    const double delta_t = air_t - surf_t;
    const double adt = fabs(delta_t);
    const double c3 = cbrt(adt);
    double coef = (delta_t > 0.) ? nat_pos : nat_neg;
    if (adt < 1e-3) coef = 1.31;
    if (delta_t != delta_t || nat_pos != nat_pos) {  // unreachable!() in the reference (NaN temperatures / tilt)
        bad |= FLAG_UNREACHABLE;
        coef = __builtin_nan("");
    }
    const double h = coef * c3;
    return (h < 0.1) ? 0.1 : h;  // MIN_H, convection.rs:22,105-109
}

// Forced component of the TARP exterior coefficient — reference src/convection.rs:151-168
// with roughness_index == 1 -> COEFFICIENTS[1] = 1.67 (surface.rs:618,633,649).
__device__ __forceinline__ double tarp_forced(double air_speed, double area, double perimeter, bool windward) {
    const double wf = windward ? 1.0 : 0.5;
    return 2.537 * wf * 1.67 * sqrt(perimeter * air_speed / area);
}

// is_windward — reference src/surface.rs:37-46 (sin/cos of the wind direction come from the host).
__device__ __forceinline__ bool is_windward(double sin_wd, double cos_wd, double cos_tilt, double nx, double ny) {
    return (fabs(cos_tilt) < 0.98) ? ((nx * sin_wd + ny * cos_wd) > 0.0) : true;
}

// (ir / sigma)^0.25 - 273.15 — reference src/surface.rs:647,692.
__device__ __forceinline__ double ir_to_rad_temperature(double ir) {
    return sqrt(sqrt(ir / kSigma)) - 273.15;
}

// 4 eps sigma (273.15 + (T_rad + T_surf)/2)^3 — reference src/surface.rs:941-948.
__device__ __forceinline__ double rad_hs(double emis, double rad_t, double surf_t) {
    const double tm = 273.15 + (rad_t + surf_t) / 2.;
    return 4. * emis * kSigma * ((tm * tm) * tm);
}

// ---- gas properties: reference src/gas.rs:45-74,155-179 -------------------
__device__ __forceinline__ void gas_coeffs(int gas, double &k0, double &k1, double &m0, double &m1,
                                           double &c0, double &c1, double &mass) {
    switch (gas) {
    case 1: k0 = 2.285e-3; k1 = 5.149e-5; m0 = 3.379e-6; m1 = 6.451e-8; c0 = 521.9285; c1 = 0.; mass = 39.948; break;
    case 2: k0 = 9.443e-4; k1 = 2.826e-5; m0 = 2.213e-6; m1 = 7.777e-8; c0 = 248.0907; c1 = 0.; mass = 83.8; break;
    case 3: k0 = 4.538e-4; k1 = 1.723e-5; m0 = 1.069e-6; m1 = 7.414e-8; c0 = 158.3397; c1 = 0.; mass = 131.30; break;
    default: k0 = 2.873e-3; k1 = 7.760e-5; m0 = 3.723e-6; m1 = 4.94e-8; c0 = 1002.7370; c1 = 1.2324e-2; mass = 28.97; break;
    }
}

__device__ __forceinline__ double air_density(double temp_k) {  // gas.rs:175-179
    return 101325. * 28.97 / (8314.46261815324 * temp_k);
}
__device__ __forceinline__ double air_heat_capacity(double temp_k) {  // gas.rs:49,165-167
    return 1002.7370 + 1.2324e-2 * temp_k;
}
// ThermalZone::mcp — reference src/zone.rs:59-65
__device__ __forceinline__ double zone_mcp(double volume, double temp) {
    return volume * air_density(temp + 273.15) * air_heat_capacity(temp + 273.15) / 1.;
}

// x^y for the Nusselt correlations (x > 0: Rayleigh numbers, aspect ratios; a negative x gives NaN as powf does for
// these non-integer exponents). exp(y ln x) has a relative error of about |y ln x| * 1e-16 (<= 5e-15 here), far inside
// the 1e-9 parity tolerance, at less than half the instructions of the correctly rounded pow().
__device__ __forceinline__ double powr(double x, double y) { return exp(y * log(x)); }

// nu_90 — reference src/gas.rs:285-307
__device__ __attribute__((noinline)) double nu_90(double ra, double a_gi, int &bad) {
    double nu1;
    if (ra <= 1e4) {
        nu1 = 1. + 1.7596678 * 1e-10 * powr(ra, 2.2984755);
    } else if (ra < 5e4) {
        nu1 = 0.028154 * powr(ra, 0.4134);
    } else if (ra > 5e4) {
        nu1 = 0.0673838 * cbrt(ra);
    } else {
        bad |= FLAG_UNREACHABLE;
        nu1 = __builtin_nan("");
    }
    const double nu2 = 0.242 * powr(ra / a_gi, 0.272);
    return (nu1 > nu2) ? nu1 : nu2;
}

// nu_60 — reference src/gas.rs:249-263
__device__ __attribute__((noinline)) double nu_60(double ra, double a_gi) {
    const double g = 0.5 / powr(1. + powr(ra / 3160., 20.6), 0.1);
    const double t = 0.0936 * powr(ra, 0.314) / (1. + g);
    const double t2 = t * t, t4 = t2 * t2;
    const double t7 = (t * t2) * t4;  // powi(7) as compiler-rt expands it
    const double nu1 = powr(1. + t7, 1. / 7.);
    const double nu2 = (0.104 + 0.175 / a_gi) * powr(ra, 0.283);
    return (nu1 > nu2) ? nu1 : nu2;
}

// nusselt — reference src/gas.rs:197-315
__device__ __attribute__((noinline)) double nusselt(double ra, double gamma, double a_gi, int &bad) {
    const double THIRTY_RAD = 30. * kPi / 180.;
    const double EPSILON_RAD = 0.5 * kPi / 180.;
    gamma = fmod(gamma, kPi);
    if (gamma >= 0.0 && gamma < 2. * THIRTY_RAD - EPSILON_RAD) {  // nu_0_60, gas.rs:227-244
        const double cos_gamma = cos(gamma);
        const double x = 1. - 1708. / (ra * cos_gamma);
        const double a = (x + fabs(x)) / 2.;
        const double b = 1. - 1708. * powr(sin(1.8 * gamma), 1.6) / (ra * cos_gamma);
        const double c = cbrt(ra * cos_gamma / 5830.) - 1.;
        return 1. + 1.44 * a * b + (c + fabs(c)) / 2.;
    } else if (gamma < 2. * THIRTY_RAD + EPSILON_RAD) {
        return nu_60(ra, a_gi);
    } else if (gamma < 3. * THIRTY_RAD - EPSILON_RAD) {  // nu_60_90, gas.rs:269-280
        const double nu60 = nu_60(ra, a_gi);
        const double nu90 = nu_90(ra, a_gi, bad);
        const double x = (gamma - kPi / 3.) / (kPi / 2. - kPi / 3.);
        return nu60 + (nu90 - nu60) * x;
    } else if (gamma < 3. * THIRTY_RAD + EPSILON_RAD) {
        return nu_90(ra, a_gi, bad);
    } else if (gamma < 6. * THIRTY_RAD) {  // nu_90_180, gas.rs:312-315
        const double nu_v = nu_90(ra, a_gi, bad);
        return 1. + (nu_v - 1.) * sin(gamma);
    }
    bad |= FLAG_UNREACHABLE;
    return __builtin_nan("");
}

// Cavity::u_value + Gas::cavity_convection + Gas::raleigh —
// reference src/cavity.rs:59-69, src/gas.rs:82-152.
__device__ inline double cavity_u_value(const CavityDev &c, double t_front, double t_back, int &bad) {
    double k0, k1, m0, m1, c0, c1, mass;
    gas_coeffs(c.gas, k0, k1, m0, m1, c0, c1, mass);
    double gamma = c.angle;
    if (t_front > t_back) gamma = 180. * (kPi / 180.) - gamma;  // (180.).to_radians() - gamma
    const double a_gi = c.height / c.thickness;
    const double temp = ((t_front + 273.15) + (t_back + 273.15)) / 2.;
    const double lambda = k0 + k1 * temp;
    double ra;
    if (fabs(t_front - t_back) < 1e-10) {
        ra = 0.0000001;
    } else {
        const double beta = 1. / temp;
        const double c_p = (c1 != 0.) ? c0 + c1 * temp : c0;
        const double mu = m0 + m1 * temp;
        const double rho = 101325. * mass / (8314.46261815324 * temp);
        const double th = c.thickness;
        ra = (rho * rho) * ((th * th) * th) * 9.81 * beta * c_p * fabs(t_front - t_back) / (mu * lambda);
    }
    const double nu = nusselt(ra, gamma, a_gi, bad);
    const double conv = nu * lambda / c.thickness;
    const double tm = (t_back + t_front) / 2. + 273.15;
    const double rad = 4. * ((tm * tm) * tm) * kSigma * c.ein * c.eout / (1. - (1. - c.ein) * (1. - c.eout));
    return rad + conv;
}

}  // namespace heat
