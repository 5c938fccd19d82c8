//! Raw bindings to `libptcore.so`: every type and entry point of `include/ptcore.h`, field for field and
//! in the header's order.  No Rust toolchain exists where this library is built, so this file has not been
//! through rustc; `tests/test_rust_binding.py` holds its structs, constants and function signatures
//! against the header instead.
//!
//! The seam in the reference is `Camera::render(&self, objects, lights, buffer)` (`src/camera.rs:162-167`);
//! `../render_hip.rs` is the module that fills these descriptors from the reference's own types.
#![allow(non_camel_case_types)]

use std::os::raw::c_char;

pub const PT_OK: i32 = 0;
pub const PT_ERR_INVALID: i32 = 1;
pub const PT_ERR_HIP: i32 = 2;
pub const PT_ERR_NOMEM: i32 = 3;
pub const PT_ERR_UNSUPPORTED: i32 = 4;
pub const PT_ERR_STATE: i32 = 5;

pub const PT_PRIM_TRI: i32 = 0;
pub const PT_PRIM_SPHERE: i32 = 1;
pub const PT_PRIM_QUAD: i32 = 2;
pub const PT_PRIM_INSTANCE: i32 = 3;
pub const PT_PRIM_MOVING_SPHERE: i32 = 4;

pub const PT_NODE_LEAF: u32 = 0x8000_0000;

pub const PTS_TEX_SOLID: u32 = 0;
pub const PTS_TEX_CHECKERED: u32 = 1;
pub const PTS_TEX_IMAGE: u32 = 2;
pub const PTS_TEX_NOISE: u32 = 3;

pub const PTS_MAT_LAMBERTIAN: u32 = 0;
pub const PTS_MAT_METAL: u32 = 1;
pub const PTS_MAT_DIELECTRIC: u32 = 2;
pub const PTS_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const PTS_MAT_DISNEY: u32 = 4;
pub const PTS_MAT_DISNEY_DIFFUSE: u32 = 5;
pub const PTS_MAT_DISNEY_METAL: u32 = 6;
pub const PTS_MAT_DISNEY_GLASS: u32 = 7;
pub const PTS_MAT_DISNEY_CLEARCOAT: u32 = 8;
pub const PTS_MAT_DISNEY_SHEEN: u32 = 9;
pub const PTS_MAT_COUNT: u32 = 10;

pub const PTS_MAX_MEDIA: u32 = 8;
pub const PTS_MAX_MOVING: u32 = 8;

pub const PT_OPT_BSDF_SAMPLE_PROBABILITY: i32 = 0;
pub const PT_OPT_MAX_PATHS: i32 = 1;
pub const PT_OPT_PROFILE: i32 = 2;
pub const PT_OPT_COUNTERS: i32 = 3;
pub const PT_OPT_TRAVERSAL: i32 = 4;
pub const PT_OPT_PIPES: i32 = 5;
pub const PT_OPT_REFILL_BELOW: i32 = 6;
pub const PT_OPT_LEAF_BATCH: i32 = 7;
pub const PT_OPT_BLOCKS_PER_CU: i32 = 8;
pub const PT_OPT_TAIL_BELOW: i32 = 9;
pub const PT_OPT_ROLLING: i32 = 10;
pub const PT_OPT_DEVICE_BVH: i32 = 11;
pub const PT_OPT_INTEGRATOR: i32 = 12;
pub const PT_OPT_LIVE_FACTOR: i32 = 13;
pub const PT_OPT_PACKET: i32 = 14;
pub const PT_OPT_BATCH_TICKS: i32 = 15;
pub const PT_OPT_SHADE_CLAIM: i32 = 16;

/// One `FlatBVHNode` (`src/bvh.rs:667-679`) in 32 bytes.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pt_node {
    pub bmin: [f32; 3],
    pub bmax: [f32; 3],
    pub a: u32,
    pub b: u32,
}

/// One `InnerPrimitiveBuffer<T>` (`src/objects.rs:105-109`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pt_buffer_desc {
    pub prim_first: u32,
    pub prim_count: u32,
    pub node_first: u32,
    pub node_count: u32,
}

/// One `PrimitiveBuffer` (`src/objects.rs:130-136`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pt_primset_desc {
    pub tris: pt_buffer_desc,
    pub spheres: pt_buffer_desc,
    pub quads: pt_buffer_desc,
    pub instances: pt_buffer_desc,
}

/// One `Instance` (`src/objects.rs:281-286`); matrices row-major.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct pt_instance_desc {
    pub transform: [f32; 16],
    pub inverse: [f32; 16],
    pub child_set: u32,
    /// Levels of an instance that was nested, outer to inner: records `chain_first .. chain_first + chain_count`.
    pub chain_first: u32,
    pub chain_count: u32,
    pub reserved: u32,
}

/// One `Arc<dyn Material>`.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct pt_material_desc {
    pub r#type: u32,
    pub texture: i32,
    pub p: [f32; 14],
    pub derived: [f32; 8],
}

/// One `Arc<dyn Texture>` (`src/textures.rs:49-104`).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct pts_texture {
    pub r#type: u32,
    pub color: [f32; 3],
    pub odd: i32,
    pub even: i32,
    pub scale: f32,
    pub image: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct pts_image {
    pub width: u32,
    pub height: u32,
    pub offset: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct pts_medium {
    pub center: [f32; 3],
    pub radius: f32,
    pub density: f32,
    pub texture: i32,
}

#[repr(C)]
pub struct pt_scene_desc {
    pub n_tris: u32,
    pub tri_data: *const f32,
    pub tri_mat: *const i32,
    pub n_spheres: u32,
    pub spheres: *const f32,
    pub sphere_mat: *const i32,
    pub n_quads: u32,
    pub quads: *const f32,
    pub quad_mat: *const i32,
    pub n_instances: u32,
    pub instances: *const pt_instance_desc,
    pub n_nodes: u32,
    pub nodes: *const pt_node,
    pub n_sets: u32,
    pub sets: *const pt_primset_desc,
    pub n_materials: u32,
    pub materials: *const pt_material_desc,
    pub n_textures: u32,
    pub textures: *const pts_texture,
    pub n_images: u32,
    pub images: *const pts_image,
    pub n_texel_bytes: u64,
    pub texels: *const u8,
    pub n_media: u32,
    pub media: *const pts_medium,
    pub n_moving: u32,
    pub moving: *const f32,
}

/// The fields of `Camera` that render/get_ray/ray_color read (`src/camera.rs:34-61`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pt_camera_desc {
    pub image_width: u32,
    pub image_height: u32,
    pub center: [f32; 3],
    pub pixel00: [f32; 3],
    pub pixel_du: [f32; 3],
    pub pixel_dv: [f32; 3],
    pub defocus_disk_u: [f32; 3],
    pub defocus_disk_v: [f32; 3],
    pub defocus_angle: f32,
    pub sqrt_spp: u32,
    pub inv_sqrt_spp: f32,
    pub max_depth: u32,
    pub background: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pt_hit {
    pub prim_type: i32,
    pub prim_id: i32,
    pub inst_id: i32,
    pub t: f32,
    pub u: f32,
    pub v: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pt_stats {
    pub rays: u64,
    pub paths: u64,
    pub extend_launches: u64,
    pub extend_ms: f64,
    pub extend_busy_ms: f64,
    pub shade_ms: f64,
    pub other_ms: f64,
    pub render_ms: f64,
    pub host_wait_ms: f64,
    pub node_tests: u64,
    pub tri_tests: u64,
    pub sphere_tests: u64,
    pub quad_tests: u64,
    pub device_bytes: u64,
    pub bad_refs: u64,
    pub walk_ms: f64,
    pub packet_ms: f64,
    pub walk_launches: u64,
    pub packet_launches: u64,
    pub shadow_rays: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pt_material_probe_in {
    pub w_out: [f32; 3],
    pub u: f32,
    pub v: f32,
    pub p: [f32; 3],
    pub front_face: u32,
    pub pixel: u32,
    pub stratum: u32,
    pub w_in: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pt_material_probe_out {
    pub w_in: [f32; 3],
    pub draws: u32,
    pub bsdf: [f32; 3],
    pub pdf: f32,
    pub bsdf_at: [f32; 3],
    pub pdf_at: f32,
    pub emitted: [f32; 3],
    pub reserved: u32,
}

/// Largest `n` of `pt_trace_first_hit_static`.
pub const PT_STATIC_TRACE_MAX_RAYS: u32 = 262144;

/// Opaque: one context per GPU, not re-entrant.
#[repr(C)]
pub struct pt_context {
    _private: [u8; 0],
}

/// Opaque: several GPUs of one node behind one handle, not re-entrant.
#[repr(C)]
pub struct pt_group {
    _private: [u8; 0],
}

extern "C" {
    pub fn pt_create(device: i32, out: *mut *mut pt_context) -> i32;
    pub fn pt_destroy(ctx: *mut pt_context);
    pub fn pt_last_error(ctx: *const pt_context) -> *const c_char;
    pub fn pt_set_option(ctx: *mut pt_context, key: i32, value: f64) -> i32;
    pub fn pt_upload_scene(ctx: *mut pt_context, scene: *const pt_scene_desc) -> i32;
    pub fn pt_set_camera(ctx: *mut pt_context, camera: *const pt_camera_desc) -> i32;
    pub fn pt_render(ctx: *mut pt_context, stratum_begin: u32, stratum_end: u32, seed: u64) -> i32;
    pub fn pt_sync(ctx: *mut pt_context) -> i32;
    pub fn pt_render_samples(ctx: *mut pt_context, stratum_begin: u32, stratum_end: u32, seed: u64, out: *mut f32) -> i32;
    pub fn pt_generate_rays(ctx: *mut pt_context, stratum: u32, seed: u64, rays: *mut f32) -> i32;
    pub fn pt_trace_first_hit(ctx: *mut pt_context, rays: *const f32, n: u64, out: *mut pt_hit) -> i32;
    pub fn pt_trace_first_hit_at(ctx: *mut pt_context, rays: *const f32, times: *const f32, n: u64, out: *mut pt_hit) -> i32;
    pub fn pt_trace_first_hit_static(ctx: *mut pt_context, rays_device: *const f32, n: u64, out_device: *mut pt_hit) -> i32;
    pub fn pt_light_pdf(ctx: *mut pt_context, rays: *const f32, n: u64, tris_by_solid_angle: i32, out: *mut f32) -> i32;
    pub fn pt_material_probe(ctx: *mut pt_context, material: i32, seed: u64, r#in: *const pt_material_probe_in, n: u64, out: *mut pt_material_probe_out) -> i32;
    pub fn pt_read_accum(ctx: *mut pt_context, rgb_sum: *mut f32, n_samples: *mut u32) -> i32;
    pub fn pt_read_rgb8(ctx: *mut pt_context, rgb: *mut u8) -> i32;
    pub fn pt_reset_accum(ctx: *mut pt_context) -> i32;
    pub fn pt_read_triangle_order(ctx: *mut pt_context, order: *mut u32, n: u32) -> i32;
    pub fn pt_copy_accum_to_device(ctx: *mut pt_context, dst_device: *mut std::ffi::c_void) -> i32;
    pub fn pt_set_accum_from_device(ctx: *mut pt_context, src_device: *const std::ffi::c_void, n_samples: u32) -> i32;
    pub fn pt_stats_get(ctx: *mut pt_context, out: *mut pt_stats) -> i32;
    pub fn pt_stats_reset(ctx: *mut pt_context) -> i32;
    pub fn pt_group_create(devices: *const i32, n: i32, out: *mut *mut pt_group) -> i32;
    pub fn pt_group_destroy(g: *mut pt_group);
    pub fn pt_group_last_error(g: *const pt_group) -> *const c_char;
    pub fn pt_group_size(g: *const pt_group) -> i32;
    pub fn pt_group_member(g: *mut pt_group, i: i32) -> *mut pt_context;
    pub fn pt_group_split(stratum_begin: u32, stratum_end: u32, n: i32, i: i32, begin_i: *mut u32, end_i: *mut u32);
    pub fn pt_group_set_option(g: *mut pt_group, key: i32, value: f64) -> i32;
    pub fn pt_group_upload_scene(g: *mut pt_group, scene: *const pt_scene_desc) -> i32;
    pub fn pt_group_set_camera(g: *mut pt_group, camera: *const pt_camera_desc) -> i32;
    pub fn pt_group_render(g: *mut pt_group, stratum_begin: u32, stratum_end: u32, seed: u64) -> i32;
    pub fn pt_group_sync(g: *mut pt_group) -> i32;
    pub fn pt_group_read_accum(g: *mut pt_group, rgb_sum: *mut f32, n_samples: *mut u32) -> i32;
    pub fn pt_group_read_rgb8(g: *mut pt_group, rgb: *mut u8) -> i32;
    pub fn pt_group_reset_accum(g: *mut pt_group) -> i32;
    pub fn pt_group_stats_get(g: *mut pt_group, out: *mut pt_stats) -> i32;
    pub fn pt_group_stats_reset(g: *mut pt_group) -> i32;
}
