// march_walls.cpp — a compiled host driving the path through the C ABI only (no Python, no torch):
// what a non-Rust caller of libheat_amd.so looks like. Builds a small building with the model builder
// (≙ ThermalModel::new, reference src/model.rs:215-354), creates the batch, marches caller timesteps on
// a caller-owned SimulationState (≙ ThermalModel::march, src/model.rs:359-427) and prints zone and
// surface temperatures with full precision so that tests can compare them with the oracle.
//
//   g++ -std=c++17 -Iinclude examples/march_walls.cpp -Lheat_amd/lib -lheat_amd -Wl,-rpath,'$ORIGIN' -o heat_amd/lib/march_walls
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "heat_amd_setup.h"

static void die(const char *what, int rc) {
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, heat_last_error());
    std::exit(1);
}

int main(int argc, char **argv) {
    const int n_walls = argc > 1 ? std::atoi(argv[1]) : 12;
    const int n_steps = argc > 2 ? std::atoi(argv[2]) : 10;
    const int n_per_hour = 20;

    heat_model_builder *mb = heat_model_builder_create(n_per_hour, HEAT_TERRAIN_NO_SITE_DETAILS);
    const int zone_a = heat_model_builder_add_zone(mb, 600.0);
    const int zone_b = heat_model_builder_add_zone(mb, 250.0);

    heat_layer insulation{};  // polyurethane
    insulation.thickness = 0.02; insulation.conductivity = 0.0252; insulation.density = 17.5; insulation.specific_heat = 2400.;
    insulation.front_thermal_absorbtance = 0.2; insulation.back_thermal_absorbtance = 0.2;
    insulation.front_solar_absorbtance = 0.7; insulation.back_solar_absorbtance = 0.7;
    heat_layer concrete{};
    concrete.thickness = 0.2; concrete.conductivity = 0.816; concrete.density = 1700.; concrete.specific_heat = 800.;
    concrete.front_thermal_absorbtance = 0.9; concrete.back_thermal_absorbtance = 0.9;
    concrete.front_solar_absorbtance = 0.7; concrete.back_solar_absorbtance = 0.7;

    for (int i = 0; i < n_walls; i++) {
        heat_surface_in s{};
        heat_layer layers[3];
        if (i % 3 == 0) { layers[0] = concrete; s.n_layers = 1; }
        else if (i % 3 == 1) { layers[0] = insulation; layers[1] = concrete; layers[2] = insulation; s.n_layers = 3; }
        else { layers[0] = insulation; s.n_layers = 1; }
        s.layers = layers;
        s.area = 10.0 + i;
        s.perimeter = 2.0 * (s.area / 3.0 + 3.0);
        const double az = 0.4 * i;
        s.normal[0] = std::sin(az); s.normal[1] = std::cos(az); s.normal[2] = 0.0;
        s.centroid_z = 1.5 + 3.0 * (i % 4);
        s.front_kind = (i % 5 == 4) ? HEAT_BOUNDARY_SPACE : HEAT_BOUNDARY_OUTDOOR;
        s.front_zone = zone_b;
        s.back_kind = HEAT_BOUNDARY_SPACE;
        s.back_zone = (i % 2) ? zone_b : zone_a;
        int rc = heat_model_builder_add_surface(mb, &s);
        if (rc < 0) die("heat_model_builder_add_surface", rc);
    }
    const heat_batch_desc *desc = nullptr;
    const double *initial = nullptr;
    int32_t n_sub = 0;
    int rc = heat_model_builder_finish(mb, &desc, &initial, &n_sub);
    if (rc) die("heat_model_builder_finish", rc);

    std::vector<double> state(initial, initial + desc->n_state);
    heat_batch *batch = nullptr;
    rc = heat_batch_create(desc, &batch);
    if (rc) die("heat_batch_create", rc);
    rc = heat_batch_upload_state(batch, state.data(), state.size());
    if (rc) die("heat_batch_upload_state", rc);

    const double pi = 3.14159265358979323846;
    std::vector<heat_weather> w(n_sub);
    for (int step = 0; step < n_steps; step++) {
        // what another SIMPLE module would write between two marches: solar and long-wave irradiance
        for (int64_t s = 0; s < desc->n_surfaces; s++) {
            state[desc->solar_front_slot[s]] = 50.0 * (step % 7) + 3.0 * s;
            state[desc->ir_front_slot[s]] = 5.670374419e-8 * std::pow(283.15 + step, 4);
            state[desc->ir_back_slot[s]] = 5.670374419e-8 * std::pow(295.15, 4);
        }
        for (int i = 0; i < n_sub; i++) {
            w[i].dry_bulb = 10.0 + 0.5 * step;
            w[i].wind_direction = (150.0 + 10.0 * step) * (pi / 180.0);  // wind_direction.to_radians(), model.rs:373
            w[i].wind_speed = 2.0 + 0.1 * step;
        }
        const double a0[2] = {150.0, 0.0}, b0[2] = {0.0, 0.0};  // a 150 W heater in zone A (model.rs:502-507)
        rc = heat_batch_march(batch, state.data(), state.size(), w.data(), n_sub, a0, b0);
        if (rc) die("heat_batch_march", rc);
    }
    std::printf("n_state %lld dt %.17g n_sub %d\n", (long long)desc->n_state, desc->dt, n_sub);
    for (int64_t z = 0; z < desc->n_zones; z++) std::printf("zone %lld %.17g\n", (long long)z, state[desc->zone_slot[z]]);
    for (int64_t s = 0; s < desc->n_surfaces; s++) {
        const int64_t n = desc->node_offset[s + 1] - desc->node_offset[s];
        std::printf("surface %lld %.17g %.17g %.17g %.17g\n", (long long)s, state[desc->first_node_slot[s]],
                    state[desc->first_node_slot[s] + n - 1], state[desc->hs_front_slot[s]], state[desc->flow_back_slot[s]]);
    }
    heat_batch_destroy(batch);
    heat_model_builder_destroy(mb);
    return 0;
}
