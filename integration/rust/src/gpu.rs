//! `GpuThermalModel`: the reference's `SimulationModel` contract (src/model.rs:188-428) on top of the C ABI of
//! libheat_amd.so. Setup stays the reference's own (`ThermalModel::new`); only `march` changes hands.
//!
//! This file is a MODULE OF THE `heat` CRATE (`src/gpu.rs`, behind the cargo feature `gpu`), not a crate of its own:
//! two things it must reach are not visible from outside the crate —
//!   * `crate::surface_trait::SurfaceTrait` (the module is private: `mod surface_trait;`, src/lib.rs:77), the bound of
//!     `ThermalSurfaceData<T>` (src/surface.rs:315);
//!   * `ThermalLuminaire::{parent, target_space_index}` (`pub(crate)`, src/luminaire.rs:28,31; module private,
//!     src/lib.rs:82), read by the zone terms below.
//! Everything else it touches is `pub`. It calls NO private function of the reference: the first half of
//! `calculate_zones_abc` (private, src/model.rs:489) is restated here from public pieces (`zone_terms`).
//! INTEGRATION.md §2 lists every reference item used, with its visibility.
//!
//! Unverified: not compiled in this repository (no Rust toolchain in the image; the crate's dependencies are
//! unpinned git crates). The same call sequence is exercised from Python/ctypes (tests/test_parity_gpu.py) and
//! from C++ (examples/march_walls.cpp).
use crate::gpu_ffi::*;
use std::borrow::Borrow;
use std::cell::Cell;

use crate::cavity::Cavity;
use crate::discretization::UValue;
use crate::model::ThermalModel;
use crate::surface::ThermalSurfaceData;
use crate::surface_trait::SurfaceTrait;
use crate::Float;
use calendar::Date;
use communication_protocols::{MetaOptions, SimulationModel};
use simple_model::{Boundary, Fenestration, SimpleModel, SimulationState, SimulationStateHeader, Surface};
use weather::Weather;

/// The `SimulationState` slots of a surface's scalars. `SurfaceTrait` (src/surface_trait.rs:9-164) exposes the VALUES
/// of these slots and the first / last node index; the slot NUMBERS are inherent getters of `simple_model::Surface`
/// and `simple_model::Fenestration` (the ones `add_*_state` test with `.is_none()`, src/surface_trait.rs:228-362 and
/// :440-566) — hence this small trait, implemented for both.
trait SlotIndices {
    /// [hs_front, hs_back, flow_front, flow_back, solar_front, solar_back, ir_front, ir_back]
    fn scalar_slots(&self) -> Result<[usize; 8], String>;
}
macro_rules! impl_slot_indices {
    ($t:ty) => {
        impl SlotIndices for $t {
            fn scalar_slots(&self) -> Result<[usize; 8], String> {
                let need = |o: Option<usize>, what: &str| o.ok_or(format!("surface without a {} slot", what));
                Ok([
                    need(self.front_convection_coefficient_index(), "front convection coefficient")?,
                    need(self.back_convection_coefficient_index(), "back convection coefficient")?,
                    need(self.front_convective_heat_flow_index(), "front convective heat flow")?,
                    need(self.back_convective_heat_flow_index(), "back convective heat flow")?,
                    need(self.front_incident_solar_irradiance_index(), "front solar irradiance")?,
                    need(self.back_incident_solar_irradiance_index(), "back solar irradiance")?,
                    need(self.front_ir_irradiance_index(), "front IR irradiance")?,
                    need(self.back_ir_irradiance_index(), "back IR irradiance")?,
                ])
            }
        }
    };
}
impl_slot_indices!(Surface);
impl_slot_indices!(Fenestration);

/// `Cavity::gas` is a `Gas` whose fields are private (src/gas.rs:27-43); the four gases the reference defines
/// (src/gas.rs:45-74) are told apart by their molar mass, `Gas::mass()` (src/gas.rs:170).
fn gas_id(c: &Cavity) -> Result<i32, String> {
    let m = c.gas.mass();
    for (id, mass) in [(0, 28.97), (1, 39.948), (2, 83.8), (3, 131.30)] {
        // AIR, ARGON, KRYPTON, XENON = enum heat_gas
        if (m - mass).abs() < 1e-3 {
            return Ok(id);
        }
    }
    Err(format!("gas of molar mass {} is none of AIR / ARGON / KRYPTON / XENON (src/gas.rs:45-74)", m))
}

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
        Boundary::AmbientTemperature { temperature } => (1, 0, *temperature as f64),
        Boundary::Outdoor => (2, 0, 0.0),
        Boundary::Ground => (3, 0, 0.0), // rejected by heat_batch_create (the reference panics, surface.rs:642,687)
    }
}

impl Flat {
    fn push<T: SurfaceTrait + SlotIndices + Send>(&mut self, s: &ThermalSurfaceData<T>) -> Result<(), String> {
        if self.node_offset.is_empty() {
            self.node_offset.push(0)
        }
        for (mass, u) in s.discretization.segments.iter() {
            self.mass.push(*mass as f64);
            match u {
                UValue::Solid(v) => { self.uvalue.push(*v as f64); self.seg_cavity.push(-1) }
                UValue::Back => { self.uvalue.push(0.0); self.seg_cavity.push(-1) }
                UValue::None => { self.uvalue.push(f64::NAN); self.seg_cavity.push(-1) } // HEAT_E_UVALUE_NONE at create
                UValue::Cavity(c) => {
                    self.uvalue.push(0.0);
                    self.seg_cavity.push(self.cavities.len() as i32);
                    self.cavities.push(HeatCavity {
                        thickness: c.thickness as f64, height: c.height as f64, angle: c.angle as f64,
                        eout: c.eout as f64, ein: c.ein as f64, gas: gas_id(c)?, reserved: 0,
                    });
                }
            }
        }
        self.node_offset.push(self.mass.len() as i64);
        for i in 0..s.discretization.segments.len() {
            self.front_alpha.push(s.front_alphas.get(i, 0)? as f64); // Matrix::get -> Result (surface.rs:767)
            self.back_alpha.push(s.back_alphas.get(i, 0)? as f64);
        }
        let (fk, fz, fa) = boundary(&s.front_boundary, s.front_space_index);
        let (bk, bz, ba) = boundary(&s.back_boundary, s.back_space_index);
        self.front_kind.push(fk); self.front_zone.push(fz); self.front_ambient.push(fa);
        self.back_kind.push(bk); self.back_zone.push(bz); self.back_ambient.push(ba);
        self.front_emis.push(s.front_emissivity as f64); self.back_emis.push(s.back_emissivity as f64);
        self.area.push(s.area as f64); self.perimeter.push(s.perimeter as f64); self.cos_tilt.push(s.cos_tilt as f64);
        self.nx.push(s.normal.x as f64); self.ny.push(s.normal.y as f64);
        self.wind_mod.push(s.wind_speed_modifier as f64);
        let p = &s.parent;
        self.first_node.push(SurfaceTrait::first_node_temperature_index(p) as i64); // surface_trait.rs:75
        let sl = p.scalar_slots()?;
        self.hs_f.push(sl[0] as i64); self.hs_b.push(sl[1] as i64);
        self.flow_f.push(sl[2] as i64); self.flow_b.push(sl[3] as i64);
        self.solar_f.push(sl[4] as i64); self.solar_b.push(sl[5] as i64);
        self.ir_f.push(sl[6] as i64); self.ir_b.push(sl[7] as i64);
        Ok(())
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

/// The terms of `calculate_zones_abc` that do not come from surfaces (src/model.rs:500-544): HVAC and luminaire
/// power into `a`, infiltration and ventilation into `a` and `b`. They depend only on slots OTHER modules write, so
/// they are constant over the sub-timesteps of one `march` call. The surface loop (model.rs:556-590) and the
/// capacitance (model.rs:549-552) run on the device. Same statements, same order as the reference.
fn zone_terms(cpu: &ThermalModel, model: &SimpleModel, state: &SimulationState) -> Result<(Vec<f64>, Vec<f64>), String> {
    let nzones = cpu.zones.len();
    let mut a = vec![0.0 as Float; nzones];
    let mut b = vec![0.0 as Float; nzones];
    for hvac in cpu.hvacs.iter() {
        for (target_space_index, heating_cooling) in hvac.calc_cooling_heating_power(state)? {
            a[target_space_index] += heating_cooling;
        }
    }
    for luminaire in cpu.luminaires.iter() {
        let consumption = luminaire.parent.power_consumption(state).expect("Luminaire has no Power Consumption state");
        a[luminaire.target_space_index] += consumption;
    }
    let air = crate::gas::AIR;
    for i in 0..nzones {
        let space = &model.spaces[i];
        if let Some(t_inf_inwards) = space.infiltration_temperature(state) {
            let v_inf = space.infiltration_volume(state).expect("Space has infiltration temperature but not volume");
            let cp_inf_inwards = air.heat_capacity(t_inf_inwards + 273.15);
            let rho_inf_inwards = air.density(t_inf_inwards + 273.15);
            a[i] += rho_inf_inwards * v_inf * cp_inf_inwards * t_inf_inwards;
            b[i] += rho_inf_inwards * v_inf * cp_inf_inwards;
        }
        if let Some(t_vent_inwards) = space.ventilation_temperature(state) {
            let v_vent = space.ventilation_volume(state).expect("Space has ventilation temperature but not volume");
            let cp_vent_inwards = air.heat_capacity(t_vent_inwards + 273.15);
            let rho_vent_inwards = air.density(t_vent_inwards + 273.15);
            a[i] += rho_vent_inwards * v_vent * cp_vent_inwards * t_vent_inwards;
            b[i] += rho_vent_inwards * v_vent * cp_vent_inwards;
        }
    }
    Ok((a.iter().map(|v| *v as f64).collect(), b.iter().map(|v| *v as f64).collect()))
}

pub struct GpuThermalModel {
    cpu: ThermalModel,       // the reference model: still built by ThermalModel::new (setup is unchanged)
    batch: *mut HeatBatch,
    uploaded: Cell<bool>,
    nodes_stale: Cell<bool>, // node temperatures newer on the device than in the caller's state
}

impl GpuThermalModel {
    /// The node temperatures stay on the device between marches (they are 8 n of every surface's 8 n + 32 output
    /// bytes: 320 MB over PCIe per call at a million walls of 32 nodes). A module that reads them —
    /// `SurfaceTrait::get_node_temperatures` (src/surface_trait.rs:93-115), reporting — calls this first.
    pub fn fetch_node_temperatures(&self, state: &mut SimulationState) -> Result<(), String> {
        if self.nodes_stale.get() {
            check(unsafe { heat_batch_download_outputs(self.batch, state.as_mut_ptr(), state.len(), HEAT_OUT_NODE_TEMPERATURES) })?;
            self.nodes_stale.set(false);
        }
        Ok(())
    }
}

impl SimulationModel for GpuThermalModel {
    type OutputType = Self;
    type OptionType = ();
    type AllocType = (); // scratch lives on the device, inside the batch

    fn new<M: Borrow<SimpleModel>>(meta: &MetaOptions, opt: (), model: M,
                                   state: &mut SimulationStateHeader, n: usize) -> Result<Self, String> {
        let cpu = ThermalModel::new(meta, opt, model.borrow(), state, n)?; // model.rs:215-354
        let mut f = Flat::default();
        for s in cpu.surfaces.iter() { f.push(s)? }                        // surfaces, then fenestrations:
        for s in cpu.fenestrations.iter() { f.push(s)? }                   // the order of model.rs:388-408
        for z in cpu.zones.iter() {
            // ThermalZone::volume is private (zone.rs:30); it is a copy of the Space's volume (zone.rs:43)
            f.zone_volume.push(*z.reference_space.volume().unwrap() as f64);
            f.zone_slot.push(z.reference_space.dry_bulb_temperature_index()
                              .ok_or("Space without a dry bulb temperature slot")? as i64);
        }
        let desc = f.desc(cpu.dt as f64, state.len());
        let mut batch = std::ptr::null_mut();
        check(unsafe { heat_batch_create(&desc, &mut batch) })?;
        Ok(Self { cpu, batch, uploaded: false.into(), nodes_stale: false.into() })
    }

    fn allocate_memory(&self) -> Result<(), String> { Ok(()) }

    fn march<W: Weather, M: Borrow<SimpleModel>>(&self, mut date: Date, weather: &W, model: M,
                 state: &mut SimulationState, _alloc: &mut ()) -> Result<(), String> {
        let model = model.borrow();
        if !self.uploaded.get() {   // node temperatures travel up once; later calls upload only what other
            check(unsafe { heat_batch_upload_state(self.batch, state.as_ptr(), state.len()) })?;
            self.uploaded.set(true); // modules write: irradiances and zone temperatures (heat_batch_march_ex)
        }
        // weather of every sub-timestep (model.rs:371-382)
        let mut w = Vec::with_capacity(self.cpu.dt_subdivisions);
        for _ in 0..self.cpu.dt_subdivisions {
            date.add_seconds(self.cpu.dt);
            let cw = weather.get_weather_data(date);
            w.push(HeatWeather {
                dry_bulb: cw.dry_bulb_temperature
                    .ok_or("Trying to march on Thermal Model, but dry bulb temperature was not provided")? as f64,
                wind_direction: cw.wind_direction.unwrap().to_radians() as f64,
                wind_speed: cw.wind_speed.unwrap() as f64,
            });
        }
        let (a0, b0) = zone_terms(&self.cpu, model, state)?;
        // One call = all dt_subdivisions sub-timesteps (the library keeps zone-connected clusters on the chip).
        // What comes back every call: hs, convective heat flows, zone temperatures — what other modules read between
        // marches. The node temperatures follow on demand (fetch_node_temperatures).
        check(unsafe { heat_batch_march_ex(self.batch, state.as_mut_ptr(), state.len(), w.as_ptr(), w.len() as i32,
                                           a0.as_ptr(), b0.as_ptr(),
                                           HEAT_OUT_SURFACE_SCALARS | HEAT_OUT_ZONE_TEMPERATURES) })?;
        self.nodes_stale.set(true);
        Ok(())
    }
}

impl Drop for GpuThermalModel {
    fn drop(&mut self) { unsafe { heat_batch_destroy(self.batch) } }
}
