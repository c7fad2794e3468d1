//! `extern "C"` mirror of `include/rt_mi355x.h`.  Field order and types follow the header exactly.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID_ARGUMENT: c_int = -1;
pub const RT_ERR_UNSUPPORTED: c_int = -2;
pub const RT_ERR_OUT_OF_MEMORY: c_int = -3;
pub const RT_ERR_DEVICE: c_int = -4;
pub const RT_ERR_NO_DEVICE: c_int = -5;
pub const RT_ERR_IO: c_int = -6;
pub const RT_ERR_PARSE: c_int = -7;

pub const RT_TEX_SOLID: u32 = 0;
pub const RT_TEX_CHECKER: u32 = 1;
pub const RT_TEX_IMAGE: u32 = 2;

pub const RT_MAT_LAMBERTIAN: u32 = 0;
pub const RT_MAT_METAL: u32 = 1;
pub const RT_MAT_DIELECTRIC: u32 = 2;
pub const RT_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const RT_MAT_MIX: u32 = 4;
pub const RT_MAT_TROWBRIDGE_REITZ: u32 = 5;

pub const RT_MODE_LINEAR_MEAN: u32 = 0;
pub const RT_MODE_GAMMA_PER_SAMPLE: u32 = 1;

pub const RT_CTX_TIMING: u32 = 1;
pub const RT_CTX_COUNTERS: u32 = 2;

pub const RT_PIPELINE_AUTO: u32 = 0;
pub const RT_MISS: u32 = 0xFFFF_FFFF;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_texture {
    pub kind: u32,
    pub rgb: [f32; 3],
    pub inv_scale: f32,
    pub even: i32,
    pub odd: i32,
    pub width: u32,
    pub height: u32,
    pub pixels: *const u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_material {
    pub kind: u32,
    pub texture: i32,
    pub param: f32,
    pub left: i32,
    pub right: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_mesh {
    pub num_vertices: u32,
    pub positions: *const f32,
    pub num_normals: u32,
    pub normals: *const f32,
    pub num_uvs: u32,
    pub uvs: *const f32,
    pub num_faces: u32,
    pub indices: *const u32,
    pub material: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_sphere {
    pub center: [f32; 3],
    pub radius: f32,
    pub material: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_scene_desc {
    pub num_meshes: u32,
    pub meshes: *const rt_mesh,
    pub num_spheres: u32,
    pub spheres: *const rt_sphere,
    pub num_materials: u32,
    pub materials: *const rt_material,
    pub num_textures: u32,
    pub textures: *const rt_texture,
    pub spheres_first: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_scene_info {
    pub num_triangles: u32,
    pub num_spheres: u32,
    pub num_nodes: u32,
    pub depth: u32,
    pub bounds_min: [f32; 3],
    pub bounds_max: [f32; 3],
    pub device_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_camera_settings {
    pub y_fov: f32,
    pub z_near: f32,
    pub z_far: f32,
    pub transform: [f32; 12],
    pub focus_dist: f32,
    pub defocus_angle: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_camera {
    pub center: [f32; 3],
    pub pixel_00_loc: [f32; 3],
    pub pixel_delta_u: [f32; 3],
    pub pixel_delta_v: [f32; 3],
    pub defocus_disk_u: [f32; 3],
    pub defocus_disk_v: [f32; 3],
    pub defocus_angle: f32,
    pub z_near: f32,
    pub z_far: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_settings {
    pub image_width: u32,
    pub image_height: u32,
    pub max_depth: u32,
    pub samples_per_pixel: u32,
    pub background_color: [f32; 3],
    pub mode: u32,
    pub seed: u64,
    pub sample_begin: u32,
    pub sample_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_ray {
    pub origin: [f32; 3],
    pub direction: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_hit {
    pub t: f32,
    pub prim_id: u32,
    pub u: f32,
    pub v: f32,
    pub normal: [f32; 3],
    pub front_facing: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_stats {
    pub rays_traced: u64,
    pub paths: u64,
    pub node_visits: u64,
    pub triangle_tests: u64,
    pub sphere_tests: u64,
    pub extend_launches: u32,
    pub shade_launches: u32,
    pub total_ms: f64,
    pub generate_ms: f64,
    pub extend_ms: f64,
    pub shade_ms: f64,
    pub accumulate_ms: f64,
    pub timed_extend_rays: u64,
    pub timed_extend_launches: u32,
    pub pipeline: u32,
    pub bounce_ms: f64,
    pub timed_bounce_rays: u64,
    pub timed_bounce_launches: u32,
    pub bounce_launches: u32,
    pub first_bounce_ms: f64,
    pub first_bounce_rays: u64,
    pub first_bounce_launches: u32,
    pub reserved: u32,
}

pub enum rt_context {}
pub enum rt_scene {}
pub enum rt_gltf {}

pub const RT_NOALLOC_MAX_RAYS: usize = 65536;

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_device_count() -> c_int;
    pub fn rt_context_create(device: c_int, stream: *mut c_void, flags: u32, out: *mut *mut rt_context) -> c_int;
    pub fn rt_context_destroy(ctx: *mut rt_context) -> c_int;
    pub fn rt_context_synchronize(ctx: *mut rt_context) -> c_int;
    pub fn rt_context_set_max_paths(ctx: *mut rt_context, max_paths: u64) -> c_int;
    pub fn rt_context_set_pipeline(ctx: *mut rt_context, pipeline: u32) -> c_int;
    pub fn rt_context_set_timed_bounces(ctx: *mut rt_context, bounces: u32) -> c_int;
    pub fn rt_scene_create(ctx: *mut rt_context, desc: *const rt_scene_desc, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene) -> c_int;
    pub fn rt_scene_get_info(scene: *const rt_scene, out: *mut rt_scene_info) -> c_int;
    pub fn rt_scene_export_bvh(scene: *const rt_scene, boxes: *mut f32, leaf_prim: *mut i32, skip: *mut u32) -> c_int;
    pub fn rt_camera_settings_default(out: *mut rt_camera_settings) -> c_int;
    pub fn rt_camera_from_settings(settings: *const rt_camera_settings, width: u32, height: u32, out: *mut rt_camera) -> c_int;
    pub fn rt_trace_closest(ctx: *mut rt_context, scene: *const rt_scene, rays: *const rt_ray, n: usize, t_min: f32,
                            t_max: f32, use_bvh: c_int, out: *mut rt_hit, stats: *mut rt_stats) -> c_int;
    pub fn rt_trace_closest_device(ctx: *mut rt_context, scene: *const rt_scene, d_rays: *const rt_ray, n: usize,
                                   t_min: f32, t_max: f32, use_bvh: c_int, d_out: *mut rt_hit, stats: *mut rt_stats) -> c_int;
    /// Allocation-free variant of `rt_trace_closest_device` (at most `RT_NOALLOC_MAX_RAYS` rays per call).
    pub fn rt_trace_closest_noalloc(ctx: *mut rt_context, scene: *const rt_scene, d_rays: *const rt_ray, n: usize,
                                    t_min: f32, t_max: f32, use_bvh: c_int, d_out: *mut rt_hit) -> c_int;
    pub fn rt_render(ctx: *mut rt_context, scene: *const rt_scene, camera: *const rt_camera, settings: *const rt_settings,
                     accum: *mut f32, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_device(ctx: *mut rt_context, scene: *const rt_scene, camera: *const rt_camera,
                            settings: *const rt_settings, d_accum: *mut f32, stats: *mut rt_stats) -> c_int;
    /// A zeroed accumulator of 3 * width * height floats in device memory (the `d_accum` of `rt_render_device`).
    pub fn rt_accum_create(ctx: *mut rt_context, width: u32, height: u32, d_accum: *mut *mut f32) -> c_int;
    pub fn rt_accum_download(ctx: *mut rt_context, d_accum: *const f32, count: usize, host: *mut f32) -> c_int;
    pub fn rt_accum_destroy(ctx: *mut rt_context, d_accum: *mut f32) -> c_int;
    pub fn rt_allreduce_sum(ctxs: *const *mut rt_context, n_gpus: c_int, d_accum: *const *mut f32, count: usize) -> c_int;
    pub fn rt_render_node(ctxs: *const *mut rt_context, scenes: *const *const rt_scene, n_gpus: c_int,
                          camera: *const rt_camera, settings: *const rt_settings, accum: *mut f32) -> c_int;
    pub fn rt_gltf_load(path: *const c_char, out: *mut *mut rt_gltf) -> c_int;
    pub fn rt_gltf_free(gltf: *mut rt_gltf) -> c_int;
    pub fn rt_gltf_scene(gltf: *const rt_gltf) -> *const rt_scene_desc;
    pub fn rt_gltf_camera_count(gltf: *const rt_gltf) -> u32;
    pub fn rt_gltf_camera(gltf: *const rt_gltf, index: u32, out: *mut rt_camera_settings) -> c_int;
    pub fn rt_image_decode(bytes: *const c_void, size: usize, width: *mut u32, height: *mut u32, pixels: *mut *mut u8) -> c_int;
    pub fn rt_image_free(pixels: *mut u8);
}
