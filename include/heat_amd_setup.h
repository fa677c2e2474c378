/*
 * heat_amd_setup.h — setup-time half of the path: what ThermalModel::new (src/model.rs:215-354) derives
 * from constructions, flattened into the heat_batch_desc that heat_batch_create takes.
 *
 * Host code only (no GPU needed). Each function names the reference function it restates
 * (paths relative to the reference repository root).
 */
#ifndef HEAT_AMD_SETUP_H
#define HEAT_AMD_SETUP_H

#include "heat_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* simple_model::TerrainClass as used by wind_speed_modifier (src/surface.rs:150-158). */
enum heat_terrain {
    HEAT_TERRAIN_NO_SITE_DETAILS = -1, /* site_details == None -> Urban values (surface.rs:159-163) */
    HEAT_TERRAIN_COUNTRY = 0,
    HEAT_TERRAIN_SUBURBS = 1,
    HEAT_TERRAIN_CITY = 2,
    HEAT_TERRAIN_OCEAN = 3,
    HEAT_TERRAIN_URBAN = 4,
    HEAT_TERRAIN_UNSET = 5 /* Some(details) without a terrain: alpha = delta = 0 (surface.rs:142-158) */
};

/* One layer of a Construction: Material thickness + its Substance (Normal or Gas). The defaults the
 * reference applies when a property is absent (0.84 for thermal and solar absorptances, 0 for the
 * solar transmittance: surface.rs:449-461, glazing.rs:86-88, discretization.rs:265-274) are the caller's
 * to fill in. */
typedef struct heat_layer {
    int32_t is_gas;  /* Substance::Gas */
    int32_t gas;     /* enum heat_gas */
    double thickness;
    double conductivity, density, specific_heat; /* Normal substance */
    double front_thermal_absorbtance, back_thermal_absorbtance;
    double solar_transmittance;
    double front_solar_absorbtance, back_solar_absorbtance;
} heat_layer;

/* Discretization::discretize_construction (src/discretization.rs:410-544).
 * n_elements[n_layers] receives the elements per layer (0 = no-mass); returns tstep_subdivision (>= 1)
 * or a negative heat_status. */
int heat_discretize_construction(int32_t n_layers, const heat_layer *layers, double model_dt, double max_dx,
                                 double min_dt, int32_t *n_elements);

/* Number of nodes Discretization::build produces (src/discretization.rs:174-176). */
int32_t heat_count_nodes(int32_t n_layers, const int32_t *n_elements);

/* Discretization::build (src/discretization.rs:163-298): fills mass / uvalue / seg_cavity [n_nodes] and one
 * heat_cavity per gas layer (numbered from cav_base). Returns the number of cavities or a negative status. */
int heat_build_segments(int32_t n_layers, const heat_layer *layers, const int32_t *n_elements, double height,
                        double angle, double *mass, double *uvalue, int32_t *seg_cavity, heat_cavity *cavities,
                        int32_t cav_base);

/* Discretization::get_chunks (src/discretization.rs:144-160): (ini, fin) pairs. */
int heat_get_chunks(int32_t n_nodes, const double *mass, int32_t *n_massive, int32_t *massive_chunks,
                    int32_t *n_nomass, int32_t *nomass_chunks);

/* Glazing::alphas (src/glazing.rs:259-286) for layers given as (tau, rho_front, rho_back). Returns the count. */
int heat_glazing_alphas(int32_t n, const double *tau, const double *rho_front, const double *rho_back,
                        double *alphas);

/* front_alphas / back_alphas per node (src/surface.rs:463-537). Negative status where the reference panics
 * (mixture of transparent and opaque layers, gas where a solid is expected). */
int heat_node_alphas(int32_t n_layers, const heat_layer *layers, const int32_t *n_elements, int32_t n_nodes,
                     double *front_alphas, double *back_alphas);

/* wind_speed_modifier (src/surface.rs:135-166). */
double heat_wind_speed_modifier(double height, int32_t terrain);

/* ---- ThermalModel::new as a builder ------------------------------------------------------------ */
typedef struct heat_surface_in {
    const heat_layer *layers; /* the Construction, front to back */
    int32_t n_layers;
    int32_t is_fenestration; /* fenestrations are marched after surfaces (model.rs:388-408) */
    double area, perimeter;
    double normal[3];
    double centroid_z;        /* height handed to wind_speed_modifier (model.rs:272) */
    int32_t front_kind, back_kind; /* enum heat_boundary_kind */
    int32_t front_zone, back_zone;
    double front_ambient, back_ambient;
} heat_surface_in;

typedef struct heat_model_builder heat_model_builder;

/* n_per_hour: the caller's timesteps per hour (`n` of ThermalModel::new). */
heat_model_builder *heat_model_builder_create(int32_t n_per_hour, int32_t terrain);
void heat_model_builder_destroy(heat_model_builder *mb);
int heat_model_builder_add_zone(heat_model_builder *mb, double volume);
int heat_model_builder_add_surface(heat_model_builder *mb, const heat_surface_in *s);
/* Discretizes every surface (max_dx = 0.04, min_dt = 60, cavity height = 1: model.rs:236-237,252), picks
 * dt and dt_subdivisions (model.rs:261-263,326-331), registers SimulationState slots in the reference's order
 * (zones; then per surface 8 scalars + nodes: surface.rs:428-442) and exposes the result. The pointers stay
 * valid until the builder is destroyed. */
int heat_model_builder_finish(heat_model_builder *mb, const heat_batch_desc **desc, const double **initial_state,
                              int32_t *dt_subdivisions);
/* n_elements / tstep_subdivision of surface i after finish (for inspection). */
int heat_model_builder_surface_info(const heat_model_builder *mb, int64_t i, int32_t *tstep_subdivision,
                                    int32_t *n_nodes, int32_t *n_elements, int32_t n_elements_cap);

#ifdef __cplusplus
}
#endif
#endif
