//! `GpuThermalModel`: the reference's `SimulationModel` contract (src/model.rs:188-428) on top of the C ABI of
//! libheat_amd.so. Setup stays the reference's own (`ThermalModel::new`); only `march` changes hands.
//! Unverified: not compiled in this repository (see Cargo.toml). The same call sequence is exercised from
//! Python/ctypes (tests/test_parity_gpu.py) and from C++ (examples/march_walls.cpp).
pub mod ffi;
use ffi::*;
use std::borrow::Borrow;

use calendar::Date;
use communication_protocols::{MetaOptions, SimulationModel};
use heat::discretization::UValue;
use heat::model::ThermalModel;
use heat::surface::ThermalSurfaceData;
use heat::surface_trait::SurfaceTrait;
use simple_model::{Boundary, SimpleModel, SimulationState, SimulationStateHeader};
use weather::Weather;

/// The flattened `ThermalModel` ("Flattening"): owns the arrays the descriptor points into.
#[derive(Default)]
struct Flat {
    node_offset: Vec<i64>, mass: Vec<f64>, uvalue: Vec<f64>, seg_cavity: Vec<i32>,
    front_alpha: Vec<f64>, back_alpha: Vec<f64>, cavities: Vec<HeatCavity>,
    front_kind: Vec<i32>, back_kind: Vec<i32>, front_zone: Vec<i32>, back_zone: Vec<i32>,
    front_ambient: Vec<f64>, back_ambient: Vec<f64>, front_emis: Vec<f64>, back_emis: Vec<f64>,
    area: Vec<f64>, perimeter: Vec<f64>, cos_tilt: Vec<f64>, nx: Vec<f64>, ny: Vec<f64>, wind_mod: Vec<f64>,
    first_node: Vec<i64>, hs_f: Vec<i64>, hs_b: Vec<i64>, flow_f: Vec<i64>, flow_b: Vec<i64>,
    solar_f: Vec<i64>, solar_b: Vec<i64>, ir_f: Vec<i64>, ir_b: Vec<i64>,
    zone_volume: Vec<f64>, zone_slot: Vec<i64>,
}

fn boundary(b: &Boundary, space_index: Option<usize>) -> (i32, i32, f64) {
    match b {
        Boundary::Space { .. } => (0, space_index.unwrap_or(0) as i32, 0.0),
        Boundary::AmbientTemperature { temperature } => (1, 0, *temperature),
        Boundary::Outdoor => (2, 0, 0.0),
        Boundary::Ground => (3, 0, 0.0),          // rejected by heat_batch_create (the reference panics)
    }
}

impl Flat {
    fn push<T: SurfaceTrait + Send>(&mut self, s: &ThermalSurfaceData<T>) {
        if self.node_offset.is_empty() { self.node_offset.push(0) }
        for (mass, u) in s.discretization.segments.iter() {
            self.mass.push(*mass);
            match u {
                UValue::Solid(v) => { self.uvalue.push(*v); self.seg_cavity.push(-1) }
                UValue::Back => { self.uvalue.push(0.0); self.seg_cavity.push(-1) }
                UValue::None => { self.uvalue.push(f64::NAN); self.seg_cavity.push(-1) }
                UValue::Cavity(c) => {
                    self.uvalue.push(0.0);
                    self.seg_cavity.push(self.cavities.len() as i32);
                    self.cavities.push(HeatCavity { thickness: c.thickness, height: c.height, angle: c.angle,
                                                    eout: c.eout, ein: c.ein, gas: c.gas as i32, reserved: 0 });
                }
            }
        }
        self.node_offset.push(self.mass.len() as i64);
        for i in 0..s.discretization.segments.len() {
            self.front_alpha.push(s.front_alphas.get(i, 0).unwrap());
            self.back_alpha.push(s.back_alphas.get(i, 0).unwrap());
        }
        let (fk, fz, fa) = boundary(&s.front_boundary, s.front_space_index);
        let (bk, bz, ba) = boundary(&s.back_boundary, s.back_space_index);
        self.front_kind.push(fk); self.front_zone.push(fz); self.front_ambient.push(fa);
        self.back_kind.push(bk); self.back_zone.push(bz); self.back_ambient.push(ba);
        self.front_emis.push(s.front_emissivity); self.back_emis.push(s.back_emissivity);
        self.area.push(s.area); self.perimeter.push(s.perimeter); self.cos_tilt.push(s.cos_tilt);
        self.nx.push(s.normal.x); self.ny.push(s.normal.y); self.wind_mod.push(s.wind_speed_modifier);
        let p = &s.parent;
        self.first_node.push(p.first_node_temperature_index() as i64);
        self.hs_f.push(p.front_convection_coefficient_index().unwrap() as i64);
        self.hs_b.push(p.back_convection_coefficient_index().unwrap() as i64);
        self.flow_f.push(p.front_convective_heat_flow_index().unwrap() as i64);
        self.flow_b.push(p.back_convective_heat_flow_index().unwrap() as i64);
        self.solar_f.push(p.front_solar_irradiance_index().unwrap() as i64);
        self.solar_b.push(p.back_solar_irradiance_index().unwrap() as i64);
        self.ir_f.push(p.front_ir_irradiance_index().unwrap() as i64);
        self.ir_b.push(p.back_ir_irradiance_index().unwrap() as i64);
    }

    fn desc(&self, dt: f64, n_state: usize) -> HeatBatchDesc {
        HeatBatchDesc {
            abi_version: 1, reserved: 0,
            n_surfaces: self.area.len() as i64, n_zones: self.zone_volume.len() as i64,
            n_cavities: self.cavities.len() as i64, n_state: n_state as i64, dt,
            node_offset: self.node_offset.as_ptr(), mass: self.mass.as_ptr(), uvalue: self.uvalue.as_ptr(),
            seg_cavity: self.seg_cavity.as_ptr(), front_alpha: self.front_alpha.as_ptr(),
            back_alpha: self.back_alpha.as_ptr(), cavities: self.cavities.as_ptr(),
            front_kind: self.front_kind.as_ptr(), back_kind: self.back_kind.as_ptr(),
            front_zone: self.front_zone.as_ptr(), back_zone: self.back_zone.as_ptr(),
            front_ambient: self.front_ambient.as_ptr(), back_ambient: self.back_ambient.as_ptr(),
            front_emissivity: self.front_emis.as_ptr(), back_emissivity: self.back_emis.as_ptr(),
            area: self.area.as_ptr(), perimeter: self.perimeter.as_ptr(), cos_tilt: self.cos_tilt.as_ptr(),
            normal_x: self.nx.as_ptr(), normal_y: self.ny.as_ptr(), wind_modifier: self.wind_mod.as_ptr(),
            front_hs_fix: std::ptr::null(), back_hs_fix: std::ptr::null(),
            first_node_slot: self.first_node.as_ptr(), hs_front_slot: self.hs_f.as_ptr(),
            hs_back_slot: self.hs_b.as_ptr(), flow_front_slot: self.flow_f.as_ptr(),
            flow_back_slot: self.flow_b.as_ptr(), solar_front_slot: self.solar_f.as_ptr(),
            solar_back_slot: self.solar_b.as_ptr(), ir_front_slot: self.ir_f.as_ptr(),
            ir_back_slot: self.ir_b.as_ptr(), zone_volume: self.zone_volume.as_ptr(),
            zone_slot: self.zone_slot.as_ptr(),
        }
    }
}

pub struct GpuThermalModel {
    cpu: ThermalModel,            // the reference model: still built by ThermalModel::new (setup is unchanged)
    batch: *mut HeatBatch,
    uploaded: std::cell::Cell<bool>,
}

impl SimulationModel for GpuThermalModel {
    type OutputType = Self;
    type OptionType = ();
    type AllocType = ();          // scratch lives on the device, inside the batch

    fn new<M: Borrow<SimpleModel>>(meta: &MetaOptions, opt: (), model: M,
                                   state: &mut SimulationStateHeader, n: usize) -> Result<Self, String> {
        let cpu = ThermalModel::new(meta, opt, model.borrow(), state, n)?;      // model.rs:215-354
        let mut f = Flat::default();
        for s in cpu.surfaces.iter() { f.push(s) }                              // surfaces, then fenestrations:
        for s in cpu.fenestrations.iter() { f.push(s) }                         // the order of model.rs:388-408
        for z in cpu.zones.iter() {
            f.zone_volume.push(z.volume);
            f.zone_slot.push(z.reference_space.dry_bulb_temperature_index().unwrap() as i64);
        }
        let desc = f.desc(cpu.dt, state.len());
        let mut batch = std::ptr::null_mut();
        check(unsafe { heat_batch_create(&desc, &mut batch) })?;
        Ok(Self { cpu, batch, uploaded: false.into() })
    }

    fn allocate_memory(&self) -> Result<(), String> { Ok(()) }

    fn march<W: Weather, M: Borrow<SimpleModel>>(&self, mut date: Date, weather: &W, model: M,
                 state: &mut SimulationState, _alloc: &mut ()) -> Result<(), String> {
        let model = model.borrow();
        if !self.uploaded.get() {       // node temperatures travel once; later calls upload only what other
            check(unsafe { heat_batch_upload_state(self.batch, state.as_ptr(), state.len()) })?;
            self.uploaded.set(true);    // modules write: irradiances and zone temperatures (heat_batch_march)
        }
        // weather of every sub-timestep (model.rs:371-382)
        let mut w = Vec::with_capacity(self.cpu.dt_subdivisions);
        for _ in 0..self.cpu.dt_subdivisions {
            date.add_seconds(self.cpu.dt);
            let cw = weather.get_weather_data(date);
            w.push(HeatWeather {
                dry_bulb: cw.dry_bulb_temperature
                    .ok_or("Trying to march on Thermal Model, but dry bulb temperature was not provided")?,
                wind_direction: cw.wind_direction.unwrap().to_radians(),
                wind_speed: cw.wind_speed.unwrap() });
        }
        // host-side zone terms (HVAC, luminaires, infiltration, ventilation: model.rs:500-544): the first half of
        // calculate_zones_abc, before the surface loop — a small refactor of the reference exposes it
        let (a0, b0) = self.cpu.zone_terms_without_surfaces(model, state)?;
        // one call = all dt_subdivisions sub-timesteps: the library keeps zone-connected clusters on the chip
        check(unsafe { heat_batch_march(self.batch, state.as_mut_ptr(), state.len(),
                                        w.as_ptr(), w.len() as i32, a0.as_ptr(), b0.as_ptr()) })
    }
}

impl Drop for GpuThermalModel { fn drop(&mut self) { unsafe { heat_batch_destroy(self.batch) } } }
