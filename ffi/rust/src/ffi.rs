//! `#[repr(C)]` mirrors of include/mirt.h, field for field, and the entry points.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const MIRT_OK: c_int = 0;
pub const MIRT_PRIM_SPHERE: u32 = 0;
pub const MIRT_PRIM_MOVING_SPHERE: u32 = 1;
pub const MIRT_PRIM_RECT: u32 = 2;
pub const MIRT_PRIM_BOX: u32 = 3;
pub const MIRT_PRIM_FLAG_BOUNDARY: u32 = 1;
pub const MIRT_RECT_XY: u32 = 0;
pub const MIRT_RECT_XZ: u32 = 1;
pub const MIRT_RECT_YZ: u32 = 2;
pub const MIRT_OP_TRANSLATE: u32 = 0;
pub const MIRT_OP_ROTATE_Y: u32 = 1;
pub const MIRT_OP_FLIP_FACE: u32 = 2;
pub const MIRT_MAX_CHAIN_OPS: usize = 4;
pub const MIRT_MAX_MEDIA: usize = 8;
pub const MIRT_MAT_LAMBERTIAN: u32 = 0;
pub const MIRT_MAT_METAL: u32 = 1;
pub const MIRT_MAT_DIELECTRIC: u32 = 2;
pub const MIRT_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const MIRT_MAT_ISOTROPIC: u32 = 4;
pub const MIRT_TEX_SOLID: u32 = 0;
pub const MIRT_TEX_CHECKER: u32 = 1;
pub const MIRT_TEX_NOISE: u32 = 2;
pub const MIRT_TEX_IMAGE: u32 = 3;
pub const MIRT_LIGHT_XZ_RECT: u32 = 0;
pub const MIRT_LIGHT_SPHERE: u32 = 1;
pub const MIRT_LIGHT_OTHER: u32 = 2;
pub const MIRT_LIGHTS_NONE: u32 = 0;
pub const MIRT_LIGHTS_OBJECT: u32 = 1;
pub const MIRT_LIGHTS_WORLD: u32 = 2;
pub const MIRT_BUILD_AUTO: u32 = 0;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_prim_header { pub material: u32, pub chain: u32, pub order: u32, pub type_index: u32, pub object_index: u32, pub flags: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_sphere { pub h: mirt_prim_header, pub center: [f32; 3], pub radius: f32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_moving_sphere { pub h: mirt_prim_header, pub center0: [f32; 3], pub center1: [f32; 3], pub time0: f32, pub time1: f32, pub radius: f32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_rect { pub h: mirt_prim_header, pub axis: u32, pub a0: f32, pub a1: f32, pub b0: f32, pub b1: f32, pub k: f32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_box { pub h: mirt_prim_header, pub min: [f32; 3], pub max: [f32; 3] }
#[repr(C)] #[derive(Clone, Copy, Default, PartialEq)]
pub struct mirt_chain_op { pub kind: u32, pub p: [f32; 3] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_chain { pub first_op: u32, pub n_ops: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_medium { pub boundary_tag: u32, pub boundary_index: u32, pub neg_inv_density: f32, pub material: u32,
                         pub chain: u32, pub order: u32, pub type_index: u32, pub object_index: u32 }
#[repr(C)] #[derive(Clone, Copy, Default, PartialEq)]
pub struct mirt_material { pub kind: u32, pub texture: u32, pub param: f32 }
#[repr(C)] #[derive(Clone, Copy, Default, PartialEq)]
pub struct mirt_texture { pub kind: u32, pub rgb: [f32; 3], pub odd: u32, pub even: u32, pub scale: f32, pub index: u32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct mirt_perlin { pub random_vec: [[f32; 3]; 256], pub perm_x: [u32; 256], pub perm_y: [u32; 256], pub perm_z: [u32; 256] }
#[repr(C)] #[derive(Clone, Copy)]
pub struct mirt_image { pub rgb8: *const u8, pub width: u32, pub height: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_light { pub kind: u32, pub p: [f32; 5] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_light_group { pub first_light: u32, pub n_lights: u32 }

#[repr(C)]
pub struct mirt_scene_desc {
    pub spheres: *const mirt_sphere, pub n_spheres: u32,
    pub moving_spheres: *const mirt_moving_sphere, pub n_moving_spheres: u32,
    pub rects: *const mirt_rect, pub n_rects: u32,
    pub boxes: *const mirt_box, pub n_boxes: u32,
    pub chains: *const mirt_chain, pub n_chains: u32,
    pub chain_ops: *const mirt_chain_op, pub n_chain_ops: u32,
    pub media: *const mirt_medium, pub n_media: u32,
    pub materials: *const mirt_material, pub n_materials: u32,
    pub textures: *const mirt_texture, pub n_textures: u32,
    pub perlins: *const mirt_perlin, pub n_perlins: u32,
    pub images: *const mirt_image, pub n_images: u32,
    pub lights_mode: u32,
    pub lights: *const mirt_light, pub n_lights: u32,
    pub light_groups: *const mirt_light_group, pub n_light_groups: u32,
    pub time0: f32, pub time1: f32,
    pub build_flags: u32,
}

/// The ten fields of src/camera.rs:4-15.
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_camera { pub origin: [f32; 3], pub lower_left_corner: [f32; 3], pub horizontal: [f32; 3], pub vertical: [f32; 3],
                         pub u: [f32; 3], pub v: [f32; 3], pub w: [f32; 3], pub lens_radius: f32, pub time0: f32, pub time1: f32 }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_render_params { pub width: u32, pub height: u32, pub samples_per_pixel: u32, pub first_sample: u32, pub max_depth: u32,
                                pub background: [f32; 3], pub seed: u64, pub path_pool: u32, pub flags: u32,
                                pub first_row: u32, pub n_rows: u32 }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_stats { pub render_ms: f64, pub samples: u64, pub segments: u64, pub nodes_visited: u64, pub prims_tested: u64,
                        pub iterations: u32, pub path_pool: u32, pub kernel_ms: [f64; 4], pub kernel_launches: [u32; 4],
                        pub phase_cycles: [u64; 6], pub node_wave_steps: u64, pub prim_wave_steps: u64,
                        pub scene_in_lds: u32, pub lds_bytes: u32 }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_ray { pub origin: [f32; 3], pub direction: [f32; 3], pub time: f32 }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct mirt_hit_record { pub point: [f32; 3], pub t: f32, pub normal: [f32; 3], pub front_face: u32, pub u: f32, pub v: f32,
                             pub type_index: u32, pub object_index: u32, pub color: [f32; 3], pub material_kind: u32 }

pub enum mirt_context {}
pub enum mirt_scene {}
pub enum mirt_frame {}

extern "C" {
    pub fn mirt_version() -> c_int;
    pub fn mirt_context_create(device_id: c_int, out: *mut *mut mirt_context) -> c_int;
    pub fn mirt_context_create_multi(device_ids: *const c_int, n_devices: c_int, row_bands: u32, out: *mut *mut mirt_context) -> c_int;
    pub fn mirt_context_device_count(ctx: *const mirt_context) -> c_int;
    pub fn mirt_context_destroy(ctx: *mut mirt_context);
    pub fn mirt_last_error(ctx: *const mirt_context) -> *const c_char;
    pub fn mirt_scene_upload(ctx: *mut mirt_context, desc: *const mirt_scene_desc, out: *mut *mut mirt_scene) -> c_int;
    pub fn mirt_scene_destroy(ctx: *mut mirt_context, scene: *mut mirt_scene);
    pub fn mirt_render(ctx: *mut mirt_context, scene: *const mirt_scene, camera: *const mirt_camera, params: *const mirt_render_params,
                       rgb24_out: *mut u8, rgb_sum_out: *mut f32, stats: *mut mirt_stats) -> c_int;
    pub fn mirt_render_accumulate(ctx: *mut mirt_context, scene: *const mirt_scene, camera: *const mirt_camera,
                                  params: *const mirt_render_params, rgb_sum_device: *mut f32, stream: *mut c_void,
                                  stats: *mut mirt_stats) -> c_int;
    pub fn mirt_render_accumulate_static(ctx: *mut mirt_context, scene: *const mirt_scene, camera: *const mirt_camera,
                                         params: *const mirt_render_params, rgb_sum_device: *mut f32, stream: *mut c_void,
                                         stats: *mut mirt_stats) -> c_int;
    pub fn mirt_static_workspace_paths() -> c_int;
    pub fn mirt_resolve(ctx: *mut mirt_context, rgb_sum_device: *const f32, width: u32, height: u32, samples_per_pixel: u32,
                        rgb24_device: *mut u8, stream: *mut c_void) -> c_int;
    pub fn mirt_frame_create(ctx: *mut mirt_context, width: u32, height: u32, out: *mut *mut mirt_frame) -> c_int;
    pub fn mirt_frame_destroy(ctx: *mut mirt_context, frame: *mut mirt_frame);
    pub fn mirt_frame_render(ctx: *mut mirt_context, scene: *const mirt_scene, camera: *const mirt_camera,
                             params: *const mirt_render_params, frame: *mut mirt_frame, stats: *mut mirt_stats) -> c_int;
    pub fn mirt_frame_samples(frame: *const mirt_frame) -> u32;
    pub fn mirt_frame_read(ctx: *mut mirt_context, frame: *const mirt_frame, rgb24_out: *mut u8, rgb_sum_out: *mut f32) -> c_int;
    pub fn mirt_frame_save(ctx: *mut mirt_context, frame: *const mirt_frame, path: *const c_char) -> c_int;
    pub fn mirt_frame_load(ctx: *mut mirt_context, path: *const c_char, out: *mut *mut mirt_frame) -> c_int;
    pub fn mirt_trace_primary(ctx: *mut mirt_context, scene: *const mirt_scene, rays: *const mirt_ray, n: usize, seed: u64,
                              type_index: *mut u32, object_index: *mut u32, t: *mut f32) -> c_int;
    pub fn mirt_trace_records(ctx: *mut mirt_context, scene: *const mirt_scene, rays: *const mirt_ray, n: usize, seed: u64,
                              records: *mut mirt_hit_record) -> c_int;
    pub fn mirt_trace_device(ctx: *mut mirt_context, scene: *const mirt_scene, rays_device: *const mirt_ray, n: usize,
                             seed: u64, type_index_device: *mut u32, object_index_device: *mut u32, t_device: *mut f32,
                             stream: *mut c_void) -> c_int;
}
