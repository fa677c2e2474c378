//! FFI declarations of include/heat_amd.h (ABI version 1) — `src/gpu_ffi.rs` of the `heat` crate, feature `gpu`.
//! Unverified: not compiled in this repository. Struct layouts are checked against the C header by
//! tests/test_abi_symbols.py (ctypes mirrors of the same structs vs gcc's offsetof).
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct HeatCavity { pub thickness: f64, pub height: f64, pub angle: f64,
                                   pub eout: f64, pub ein: f64, pub gas: i32, pub reserved: i32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct HeatWeather { pub dry_bulb: f64, pub wind_direction: f64, pub wind_speed: f64 }
#[repr(C)] pub struct HeatBatchDesc {
    pub abi_version: i32, pub reserved: i32,
    pub n_surfaces: i64, pub n_zones: i64, pub n_cavities: i64, pub n_state: i64, pub dt: f64,
    pub node_offset: *const i64, pub mass: *const f64, pub uvalue: *const f64, pub seg_cavity: *const i32,
    pub front_alpha: *const f64, pub back_alpha: *const f64, pub cavities: *const HeatCavity,
    pub front_kind: *const i32, pub back_kind: *const i32, pub front_zone: *const i32, pub back_zone: *const i32,
    pub front_ambient: *const f64, pub back_ambient: *const f64,
    pub front_emissivity: *const f64, pub back_emissivity: *const f64,
    pub area: *const f64, pub perimeter: *const f64, pub cos_tilt: *const f64,
    pub normal_x: *const f64, pub normal_y: *const f64, pub wind_modifier: *const f64,
    pub front_hs_fix: *const f64, pub back_hs_fix: *const f64,
    pub first_node_slot: *const i64, pub hs_front_slot: *const i64, pub hs_back_slot: *const i64,
    pub flow_front_slot: *const i64, pub flow_back_slot: *const i64,
    pub solar_front_slot: *const i64, pub solar_back_slot: *const i64,
    pub ir_front_slot: *const i64, pub ir_back_slot: *const i64,
    pub zone_volume: *const f64, pub zone_slot: *const i64,
}
#[repr(C)] pub struct HeatBatchOptions {
    pub device: i32, pub force_general: i32, pub nodes_per_lane: i32, pub use_graph: i32,
    pub stream: *mut c_void, pub n_ranks: i32, pub rank: i32, pub no_palette: i32, pub no_fusion: i32,
}
#[repr(C)] pub struct HeatBatch { _private: [u8; 0] }

pub const HEAT_COMM_ID_BYTES: usize = 128;

extern "C" {
    pub fn heat_batch_create(desc: *const HeatBatchDesc, out: *mut *mut HeatBatch) -> c_int;
    pub fn heat_batch_create_ex(desc: *const HeatBatchDesc, opt: *const HeatBatchOptions,
                                out: *mut *mut HeatBatch) -> c_int;
    pub fn heat_batch_destroy(b: *mut HeatBatch);
    pub fn heat_batch_upload_state(b: *mut HeatBatch, state: *const f64, n_state: usize) -> c_int;
    pub fn heat_batch_upload_inputs(b: *mut HeatBatch, state: *const f64, n_state: usize) -> c_int;
    pub fn heat_batch_download_state(b: *mut HeatBatch, state: *mut f64, n_state: usize) -> c_int;
    pub fn heat_batch_march(b: *mut HeatBatch, state: *mut f64, n_state: usize,
                            weather: *const HeatWeather, n_sub: i32,
                            zone_a0: *const f64, zone_b0: *const f64) -> c_int;
    pub fn heat_batch_march_resident(b: *mut HeatBatch, weather: *const HeatWeather, n_sub: i32,
                                     zone_a0: *const f64, zone_b0: *const f64) -> c_int;
    pub fn heat_batch_synchronize(b: *mut HeatBatch) -> c_int;
    // multi-GPU: one process per GPU, the library owns the RCCL communicator
    pub fn heat_comm_unique_id(id: *mut u8) -> c_int;
    pub fn heat_batch_comm_init(b: *mut HeatBatch, id: *const u8) -> c_int;
    pub fn heat_comm_available() -> c_int;
    // a model cut along its zone-connected clusters: no zone shared, no communicator needed
    pub fn heat_partition(desc: *const HeatBatchDesc, n_ranks: i32, rank_of_surface: *mut i32,
                          n_shared_zones: *mut i64) -> c_int;
    pub fn heat_batch_create_shard(desc: *const HeatBatchDesc, opt: *const HeatBatchOptions,
                                   rank_of_surface: *const i32, out: *mut *mut HeatBatch) -> c_int;
    // which outputs travel back with every march (HEAT_OUT_*), and the rest on demand
    pub fn heat_batch_march_ex(b: *mut HeatBatch, state: *mut f64, n_state: usize, weather: *const HeatWeather,
                               n_sub: i32, zone_a0: *const f64, zone_b0: *const f64, what: i32) -> c_int;
    pub fn heat_batch_download_outputs(b: *mut HeatBatch, state: *mut f64, n_state: usize, what: i32) -> c_int;
    pub fn heat_batch_failed_surface(b: *const HeatBatch, index: *mut i64, kind: *mut i32) -> c_int;
    pub fn heat_batch_set_fusion(b: *mut HeatBatch, enabled: i32) -> c_int;
    pub fn heat_last_error() -> *const c_char;
}

pub const HEAT_OUT_NODE_TEMPERATURES: i32 = 1;
pub const HEAT_OUT_SURFACE_SCALARS: i32 = 2;
pub const HEAT_OUT_ZONE_TEMPERATURES: i32 = 4;
pub const HEAT_OUT_ALL: i32 = 7;

pub fn check(rc: c_int) -> Result<(), String> {
    if rc == 0 { Ok(()) } else {
        Err(unsafe { std::ffi::CStr::from_ptr(heat_last_error()) }.to_string_lossy().into_owned())
    }
}
