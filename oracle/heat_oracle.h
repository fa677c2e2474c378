/*
 * heat_oracle.h — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C (C99, f64) restatement of the wall heat-conduction hot path of
 * SIMPLE-BuildingSimulation/heat v1.0.2, written from the reference's source
 * text. It exists so that the HIP product path (heat_amd/csrc) can be checked
 * against an independent implementation of the same algorithm.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library. The product path (heat_amd/) never links, imports or
 * calls anything in oracle/.
 *
 * Parity status: the Rust reference cannot be compiled here (no cargo/rustc,
 * all dependencies are unpinned git crates) — see DESIGN.md. The oracle is
 * pinned against the reference's own known-answer tests and fixtures
 * (tests/test_oracle_*.py): test_rk4 closed form (surface.rs:1558-1620),
 * get_k_q structure tests (discretization.rs:1111-1469), get_chunks
 * (discretization.rs:1471-1558), build_* (discretization.rs:756-1058), the 15
 * Nusselt values (gas.rs:406-511), gas properties (gas.rs:334-404), ISO 9050
 * identities (glazing.rs:432-523), marching steady-state tests
 * (surface.rs:1087-1556), zones abc (model.rs:695-732) and the EnergyPlus
 * series under tests/ (validate_wall_heat_transfer.rs, validate_convection.rs).
 * The arithmetic of the external `matrix` and `polynomial` crates (source
 * absent, unpinned) is restated from their published semantics; bit-level
 * parity with those crates is unpinned.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef HEAT_ORACLE_H
#define HEAT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Boundary kinds (simple_model::Boundary as used at surface.rs:611-702). */
enum { OR_SPACE = 0, OR_AMBIENT = 1, OR_OUTDOOR = 2, OR_GROUND = 3 };

/* Gas ids (gas.rs:45-74). */
enum { OR_AIR = 0, OR_ARGON = 1, OR_KRYPTON = 2, OR_XENON = 3 };

/* Error codes (mirror the reference's Err / panic sites). */
enum {
    OR_OK = 0,
    OR_ERR_GROUND = -1,     /* surface.rs:642,687; model.rs:92 */
    OR_ERR_UVALUE_NONE = -2,/* discretization.rs:53 */
    OR_ERR_NAN_HS = 1,      /* surface.rs:704-707 */
    OR_ERR_NAN_NOMASS = 2,  /* surface.rs:850 */
    OR_ERR_NAN_ZONE = 3,    /* model.rs:417-420 */
    OR_ERR_UNREACHABLE = 4  /* convection.rs:104, gas.rs:219,296 */
};

/* One gas cavity (cavity.rs:28-50). */
typedef struct {
    double thickness;
    double height;
    double angle;
    double eout;
    double ein;
    int32_t gas;
    int32_t pad_;
} or_cavity;

/*
 * A whole thermal model as the hot path sees it: the constant data held by
 * ThermalModel / ThermalSurfaceData / Discretization (model.rs:54-77,
 * surface.rs:315-381, discretization.rs:71-87) flattened into arrays, plus the
 * slot numbers of every SimulationState element the path touches
 * (surface_trait.rs:81-125, surface.rs:428-442, zone.rs:45-50).
 * Surfaces and fenestrations are one list: the reference iterates surfaces and
 * then fenestrations (model.rs:388-408, 589-590), so fenestrations come last.
 */
typedef struct {
    int64_t n_surfaces;
    int64_t n_zones;
    int64_t n_cavities;
    double dt; /* ThermalModel::dt, model.rs:76 */

    /* Per node, CSR over surfaces: Discretization::segments (discretization.rs:73). */
    const int64_t *node_offset; /* [n_surfaces+1] */
    const double *mass;         /* segments[i].0 */
    const double *uvalue;       /* UValue::Solid(u) -> u ; UValue::Back -> 0 ; Cavity -> ignored */
    const int32_t *seg_cavity;  /* index into cavities, -1 if not a cavity; NULL = no cavities */
    const double *front_alpha;  /* surface.rs:366 */
    const double *back_alpha;   /* surface.rs:370 */
    const or_cavity *cavities;

    /* Per surface. */
    const int32_t *front_kind, *back_kind; /* surface.rs:323,329 */
    const int32_t *front_zone, *back_zone; /* surface.rs:326,332 */
    const double *front_ambient, *back_ambient; /* Boundary::AmbientTemperature{temperature} */
    const double *front_emissivity, *back_emissivity; /* surface.rs:335,338 */
    const double *area, *perimeter;       /* surface.rs:341,344 */
    const double *cos_tilt;               /* surface.rs:356 */
    const double *normal_x, *normal_y;    /* surface.rs:347 (z is not used by is_windward) */
    const double *wind_modifier;          /* surface.rs:353 */
    const double *front_hs_fix, *back_hs_fix; /* surface.rs:374-380: NaN = None; NULL = none */

    /* SimulationState slots. */
    const int64_t *first_node_slot; /* nodes are contiguous, surface_trait.rs:356-378 */
    const int64_t *hs_front_slot, *hs_back_slot;
    const int64_t *flow_front_slot, *flow_back_slot;
    const int64_t *solar_front_slot, *solar_back_slot;
    const int64_t *ir_front_slot, *ir_back_slot;

    /* Zones (zone.rs:28-56). */
    const double *zone_volume;
    const int64_t *zone_slot;
} or_model;

/* ---- leaf physics -------------------------------------------------- */
int or_is_windward(double wind_direction, double cos_tilt, double nx, double ny);
double or_wind_speed_modifier(double height, int has_site_details, int terrain /* -1 none; 0 Country 1 Suburbs 2 City 3 Ocean 4 Urban */);
double or_tarp_natural(double air_t, double surf_t, double cos_tilt, int *err);
double or_tarp_total(double air_t, double surf_t, double cos_tilt, double air_speed,
                     double area, double perimeter, int windward, int *err);

double or_gas_thermal_conductivity(int gas, double temp_k);
double or_gas_dynamic_viscosity(int gas, double temp_k);
double or_gas_heat_capacity(int gas, double temp_k);
double or_gas_mass(int gas);
double or_gas_density(int gas, double temp_k);
double or_raleigh(int gas, double t_front, double t_back, double thickness);
double or_nusselt(double ra, double gamma, double a_gi, int *err);
double or_cavity_convection(int gas, double height, double thickness, double gamma,
                            double t_front, double t_back, int *err);
double or_cavity_u_value(const or_cavity *c, double t_front, double t_back, int *err);
double or_zone_mcp(double volume, double temp);

/* ---- `matrix` crate restatements ----------------------------------- */
/* y = K x for tri-diagonal K (lo[i]=K[i][i-1], dg[i]=K[i][i], up[i]=K[i][i+1]). */
void or_prod_tri_diag(int n, const double *lo, const double *dg, const double *up,
                      const double *x, double *y);
/* Solves K x = rhs by banded Gaussian elimination without pivoting; K and rhs are consumed. */
void or_tri_diag_gaussian(int n, double *lo, double *dg, double *up, double *rhs, double *x);

/* ---- chunk-level hot path ------------------------------------------ */
void or_rearrange_k(int n, double dt, const double *c, double *lo, double *dg, double *up, double *q);
void or_rk4(int n, const double *lo, const double *dg, const double *up, const double *q, double *temps);

/* Discretization::get_k_q for chunk [ini,fin) of a surface with nrows nodes. */
int or_get_k_q(int nrows, const double *uvalue, const int32_t *seg_cavity, const or_cavity *cavities,
               int ini, int fin, const double *temperatures,
               double front_air_t, double front_rad_t, double front_hs, double front_rad_hs,
               double back_air_t, double back_rad_t, double back_hs, double back_rad_hs,
               double *lo, double *dg, double *up, double *q);

/* Discretization::get_chunks. Returns counts through n_mass/n_nomass; chunks as (ini,fin) pairs. */
void or_get_chunks(int nrows, const double *mass, int *n_mass, int *mass_chunks,
                   int *n_nomass, int *nomass_chunks);

/* ---- model-level hot path ------------------------------------------ */
/*
 * ThermalModel::march (model.rs:359-427): n_sub sub-timesteps on the flat state.
 * weather[3*i+0..2] = {dry bulb C, wind direction RADIANS, wind speed m/s} of sub-step i.
 * zone_a0/zone_b0: the host-side terms of calculate_zones_abc that precede the
 * surface loop (hvac, luminaires, infiltration, ventilation: model.rs:500-544); NULL = zeros.
 * nomass_iterations (nullable): receives the total number of passes of the
 * march_nomass loop (surface.rs:808-896) over the call.
 */
int or_model_march(const or_model *m, double *state, const double *weather, int n_sub,
                   const double *zone_a0, const double *zone_b0, int64_t *nomass_iterations);

/* iterate_surfaces (model.rs:102-180) on surfaces [s0,s1). */
int or_iterate_surfaces(const or_model *m, double *state, int64_t s0, int64_t s1,
                        double wind_direction, double wind_speed, double t_out,
                        int64_t *nomass_iterations);

/* calculate_zones_abc surface part + estimate_zones_future_temperatures. */
void or_zones_abc(const or_model *m, const double *state, double *a, double *b, double *c);

/* OpenMP-over-surfaces variant of or_model_march (the reference's disabled rayon path,
 * model.rs:113-116) — only used for the courtesy multi-core CPU baseline. */
int or_model_march_mt(const or_model *m, double *state, const double *weather, int n_sub,
                      const double *zone_a0, const double *zone_b0, int n_threads);

/* ---- setup-time restatements --------------------------------------- */
/* One construction layer as discretize_construction/build/alphas see it. */
typedef struct {
    int32_t is_gas;      /* Substance::Gas */
    int32_t gas;         /* OR_AIR.. */
    double thickness;    /* Material::thickness */
    double k, rho, cp;   /* Normal substance */
    double front_thermal_abs, back_thermal_abs; /* default 0.84 applied by caller */
    double tau;          /* solar transmittance (default 0) */
    double front_solar_abs, back_solar_abs;     /* default 0.84 applied by caller */
} or_layer;

/* Discretization::discretize_construction (discretization.rs:410-544).
 * Returns tstep_subdivision; n_elements[n_layers] filled. */
int or_discretize_construction(int n_layers, const or_layer *layers, double model_dt,
                               double max_dx, double min_dt, int *n_elements);
/* Number of nodes Discretization::build will produce (discretization.rs:174-176). */
int or_count_nodes(int n_layers, const int *n_elements);
/* Discretization::build (discretization.rs:163-298). Fills mass/uvalue/seg_cavity[n_nodes];
 * cavities_out receives one or_cavity per gas layer, numbering from cav_base. Returns #cavities or <0. */
int or_build(int n_layers, const or_layer *layers, const int *n_elements, double height, double angle,
             double *mass, double *uvalue, int32_t *seg_cavity, or_cavity *cavities_out, int cav_base);
/* Glazing::alphas over a list of (tau, rho_front, rho_back) (glazing.rs:259-286). Returns count. */
int or_glazing_alphas(int n, const double *tau, const double *rho_front, const double *rho_back, double *alphas);
/* combine (glazing.rs:215-230) exposed for the ISO 9050 identity tests. out = {tau, rho_f, rho_b, alpha_f, alpha_b}. */
void or_glazing_combine_layers(int n, const double *tau, const double *rho_front, const double *rho_back, double *out5);
/* front/back alphas per node (surface.rs:463-537). Returns 0 or <0 on the reference's panic. */
int or_node_alphas(int n_layers, const or_layer *layers, const int *n_elements, int n_nodes,
                   double *front_alphas, double *back_alphas);

#ifdef __cplusplus
}
#endif
#endif
