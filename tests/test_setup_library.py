"""The product's setup-time library (include/heat_amd_setup.h, heat_amd/csrc/setup.cpp): checked against the
reference's own known answers and, value for value, against the oracle's independent restatement. CPU only."""
import math

import numpy as np
import pytest

from heat_amd import ModelBuilder, binding, modeldict as mdl

CONCRETE = dict(k=0.816, rho=1700., cp=800.)
POLY = dict(k=0.0252, rho=17.5, cp=2400.)


def test_build_known_answers():  # discretization.rs:756-1058
    k, rho, cp, th = 1., 2.1, 1.312, 12.5 / 1000.
    d = binding.build_segments([dict(thickness=th, k=k, rho=rho, cp=cp)], [1])
    assert d["n_nodes"] == 2 and np.all(np.abs(d["mass"] - th * rho * cp / 2.) < 1e-17)
    assert abs(d["uvalue"][0] - k / th) < 1e-16 and d["uvalue"][1] == 0.0
    d = binding.build_segments([dict(thickness=th, k=k, rho=rho, cp=cp)], [0])
    assert np.all(np.abs(d["mass"]) < 1e-17) and abs(d["uvalue"][0] - k / th) < 1e-16
    solid = dict(thickness=th, k=k, rho=rho, cp=cp, front_thermal_abs=0.9, back_thermal_abs=0.8)
    gas = dict(thickness=th, is_gas=True, gas=mdl.AIR)
    d = binding.build_segments([solid, gas, solid], [1, 1, 1])
    assert d["n_nodes"] == 4 and list(d["seg_cavity"]) == [-1, 0, -1, -1]
    assert d["cavities"][0]["eout"] == 0.8 and d["cavities"][0]["ein"] == 0.9
    with pytest.raises(binding.HeatError):
        binding.build_segments([gas, solid], [0, 1])  # gas as the first layer: the reference returns Err


def test_get_chunks_known_answers():  # discretization.rs:1471-1558
    gc = binding.get_chunks
    assert gc([1.]) == ([(0, 1)], []) and gc([0.]) == ([], [(0, 1)])
    assert gc([1.] * 10) == ([(0, 10)], []) and gc([0.] * 10) == ([], [(0, 10)])
    assert gc([0., 1., 1., 0., 0.]) == ([(1, 3)], [(0, 1), (3, 5)])
    assert gc([1., 1., 1., 0., 0.]) == ([(0, 3)], [(3, 5)])


def test_iso9050_alphas():  # glazing.rs:432-523, eqs. 23-25
    import ctypes as C
    L = binding.load_library()
    tau = np.array([0.1, 0.21, 0.21]); rf = np.array([0.13, 0.1123, 0.1123]); rb = np.array([0.3, 0.34, 0.34])
    out = np.zeros(3)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert L.heat_glazing_alphas(3, dp(tau), dp(rf), dp(rb), dp(out)) == 3
    t1, t2, t3 = tau; f1, f2, f3 = rf; b1, b2, b3 = rb
    denom = (1. - b1 * f2) * (1. - b2 * f3) - t2 * t2 * b1 * f3
    a_f1, a_b1, a_f2, a_b2, a_f3 = 1 - t1 - f1, 1 - t1 - b1, 1 - t2 - f2, 1 - t2 - b2, 1 - t3 - f3
    exp = [a_f1 + (t1 * a_b1 * f2 * (1. - b2 * f3) + t1 * t2 * t2 * a_b1 * f3) / denom,
           (t1 * a_f2 * (1. - b2 * f3) + t1 * t2 * a_b2 * f3) / denom, (t1 * t2 * a_f3) / denom]
    assert np.all(np.abs(out - exp) < 1e-15)


def test_wind_speed_modifier():  # surface.rs:135-166
    L = binding.load_library()
    assert L.heat_wind_speed_modifier(0.0, -1) == 0.0
    assert L.heat_wind_speed_modifier(1.5, -1) == (270. / 10.) ** 0.14 * (1.5 / 370.) ** 0.22
    assert L.heat_wind_speed_modifier(10., 2) == (270. / 10.) ** 0.14 * (10. / 460.) ** 0.33
    assert L.heat_wind_speed_modifier(10., 5) == (270. / 10.) ** 0.14  # Some(details) without terrain: (h/0)^0 = 1


CONSTRUCTIONS = [
    [dict(thickness=0.2, **CONCRETE)],
    [dict(thickness=0.02, **POLY), dict(thickness=0.2, **CONCRETE), dict(thickness=0.02, **POLY)],
    [dict(thickness=0.02, **POLY)],
    [dict(thickness=0.1, k=0.5, rho=1200., cp=900., front_solar_abs=0.6, back_solar_abs=0.5),
     dict(thickness=0.05, is_gas=True, gas=mdl.ARGON),
     dict(thickness=0.03, k=1.0, rho=2500., cp=840., front_thermal_abs=0.9, back_thermal_abs=0.2)],
    [dict(thickness=0.003, k=1.0, rho=2500., cp=840., tau=0.8, front_solar_abs=0.1, back_solar_abs=0.12)],
    [dict(thickness=0.35, k=1.4, rho=2200., cp=1000.), dict(thickness=0.08, k=0.04, rho=30., cp=1400.),
     dict(thickness=0.012, k=0.25, rho=900., cp=1000.)],
]


@pytest.mark.parametrize("layers", CONSTRUCTIONS)
@pytest.mark.parametrize("main_dt", [180., 600., 3600., 60.])
def test_discretization_equals_the_oracles(oracle, layers, main_dt):
    a = binding.discretize(layers, main_dt, 0.04, 60., 1., 0.7)
    b = oracle.discretize(layers, main_dt, 0.04, 60., 1., 0.7)
    assert a["tstep_subdivision"] == b["tstep_subdivision"] and a["n_elements"] == b["n_elements"]
    for k in ("mass", "uvalue", "front_alpha", "back_alpha"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.array_equal(a["seg_cavity"], b["seg_cavity"]) and a["alpha_rc"] == b["alpha_rc"] == 0
    for f in ("thickness", "height", "angle", "eout", "ein", "gas"):
        assert np.array_equal(a["cavities"][f], b["cavities"][f])
    assert binding.get_chunks(a["mass"]) == oracle.get_chunks(b["mass"])


def test_translucent_double_glazing_is_rejected_like_the_reference(oracle):
    glass = dict(thickness=0.003, k=1.0, rho=2500., cp=840., tau=0.8, front_solar_abs=0.1, back_solar_abs=0.1)
    gas = dict(thickness=0.0127, is_gas=True, gas=mdl.AIR)
    d = binding.build_segments([glass, gas, glass], [0, 0, 0])
    assert d["alpha_rc"] < 0  # surface.rs:470-472 panics ("mixture of transparent and opaque layers")
    assert oracle.build_segments([glass, gas, glass], [0, 0, 0])["alpha_rc"] < 0


def test_model_builder_reproduces_config1(oracle):
    """ThermalModel::new on get_single_zone_test_building (model.rs:215-354)."""
    mb = ModelBuilder(20)
    z = mb.add_zone(600.)
    mb.add_surface([dict(thickness=0.2, front_thermal_abs=0.0, back_thermal_abs=0.0, front_solar_abs=0.0,
                         back_solar_abs=0.0, **CONCRETE)],
                   area=60., perimeter=46., normal=(0., -1., 0.), centroid_z=1.5, front_kind=mdl.OUTDOOR,
                   back_kind=mdl.SPACE, back_zone=z)
    md, state, n_sub = mb.finish()
    info = mb.surface_info(0)
    assert info == dict(tstep_subdivision=1, n_nodes=13, n_elements=[12])
    assert md["dt"] == 90.0 and n_sub == 2 and md["n_state"] == 1 + 8 + 13
    # slot order of the reference: zone, then hs f/b, flow f/b, solar f/b, ir f/b, nodes
    assert md["zone_slot"][0] == 0 and md["hs_front_slot"][0] == 1 and md["ir_back_slot"][0] == 8
    assert md["first_node_slot"][0] == 9
    assert state[0] == 22.0 and state[1] == state[2] == 1.739658084820765 and np.all(state[3:9] == 0.0)
    assert np.all(state[9:] == 22.0)
    assert md["wind_modifier"][0] == mdl.wind_speed_modifier(1.5) and md["cos_tilt"][0] == 0.0
    # identical to what the tests' helper builds from the oracle's discretization
    from test_energyplus_series import CASES, single_zone_model
    md2, st2, d2, n_sub2 = single_zone_model(oracle, *CASES["massive_no_ir_no_solar"])
    for k in ("mass", "uvalue", "front_alpha", "back_alpha", "node_offset", "first_node_slot", "area", "perimeter",
              "wind_modifier", "front_emissivity", "back_emissivity", "zone_volume", "zone_slot"):
        assert np.array_equal(md[k], md2[k]), k
    assert md["dt"] == md2["dt"] and n_sub == n_sub2 and np.array_equal(state, st2)
    mb.close()


def test_model_builder_orders_fenestrations_last_and_picks_the_strictest_dt():
    mb = ModelBuilder(6)  # 600 s main timestep
    mb.add_zone(40.)
    glass = [dict(thickness=0.003, k=1.0, rho=2500., cp=840.)]
    wall = [dict(thickness=0.02, **POLY), dict(thickness=0.2, **CONCRETE)]
    mb.add_surface(glass, 1., 4., (0., -1., 0.), 1., mdl.OUTDOOR, mdl.SPACE, is_fenestration=True)
    mb.add_surface(wall, 4., 8., (0., -1., 0.), 1., mdl.OUTDOOR, mdl.SPACE)
    md, state, n_sub = mb.finish()
    infos = [mb.surface_info(i) for i in range(2)]
    assert infos[0]["n_nodes"] > 2 and infos[1]["n_nodes"] == 2  # the wall first, the window last
    sub = max(i["tstep_subdivision"] for i in infos)
    assert n_sub == 2 * sub and md["dt"] == 3600. / (6 * sub) / 2.
    mb.close()


def test_random_constructions_equal_the_oracles():
    """tools/fuzz_setup.py: random layer stacks (1-6 layers, gas gaps, glass, random timesteps) through the product's setup
    library and the oracle's restatement, value for value — a short run of it (360 000 constructions equal in round 3)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_setup.py"), "4", "777"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and " 0 differ" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])

