//! Hand-declared binding of librt_mi355.so for the jullanggit/raytracer crate,
//! in the crate's own zero-dependency FFI idiom (compare src/mmap.rs:19-31 and
//! src/cpu_affinity.rs:9-18 of the reference).  NOT compiled in this
//! repository (no Rust toolchain here): it is the stub a maintainer would add
//! as `src/gpu.rs`, link with `-lrt_mi355`, and call from `Scene::render`.
//! Layouts mirror include/rt_mi355.h field for field.

use std::ffi::{c_char, c_int, c_void, CStr};

#[repr(C)]
pub struct RtNode { pub min: [f32; 3], pub a: u32, pub max: [f32; 3], pub b: u32 }
#[repr(C)]
pub struct RtSphere { pub center: [f32; 3], pub radius: f32, pub material: u32 }
#[repr(C)]
pub struct RtPlane { pub point: [f32; 3], pub normal: [f32; 3], pub material: u32 }
#[repr(C)]
pub struct RtTriangle {
    pub a: [f32; 3], pub e1: [f32; 3], pub e2: [f32; 3],
    pub normals_idx: u32, pub uv_idx: u32, pub bary_idx: u32, pub material: u32,
}
#[repr(C)]
pub struct RtMaterial {
    pub kind: u32, pub param: f32, pub color_kind: u32, pub rgb: [f32; 3],
    pub tex_width: u32, pub tex_height: u32, pub tex_offset: u64,
}
#[repr(C)]
pub struct RtSceneDesc {
    pub sphere_nodes: *const RtNode, pub plane_nodes: *const RtNode, pub triangle_nodes: *const RtNode,
    pub spheres: *const RtSphere, pub planes: *const RtPlane, pub triangles: *const RtTriangle,
    pub vertex_normals: *const f32, pub tex_coords: *const f32, pub bary_precomputed: *const f32,
    pub materials: *const RtMaterial, pub texels: *const u8, pub n_texel_bytes: u64,
    pub n_sphere_nodes: u32, pub n_plane_nodes: u32, pub n_triangle_nodes: u32,
    pub n_spheres: u32, pub n_planes: u32, pub n_triangles: u32,
    pub n_vertex_normals: u32, pub n_tex_coords: u32, pub n_bary: u32, pub n_materials: u32,
}
#[repr(C)]
pub struct RtRenderParams {
    pub top_left: [f32; 3], pub top_edge: [f32; 3], pub left_edge: [f32; 3], pub camera: [f32; 3],
    pub width: u32, pub height: u32, pub samples_per_pixel: u32, pub max_bounces: u32,
    pub incremental: u32, pub has_continue: u32, pub continue_sampling: u32, pub rng_mode: u32,
    pub rng_seed: u64, pub tile_x: u32, pub tile_y: u32, pub tile_w: u32, pub tile_h: u32,
}
#[repr(C)]
pub struct RtContext { _private: [u8; 0] }
#[repr(C)]
pub struct RtGroup { _private: [u8; 0] }

#[repr(C)]
pub struct RtHitRecord { pub t: f32, pub kind: u32, pub prim: u32 }
#[repr(C)]
pub struct RtStats {
    pub paths: u64, pub segments: u64, pub hits: u64, pub branch_visits: u64, pub prim_tests: [u64; 3],
    pub trace_launches: u64, pub trace_ms: f64, pub render_ms: f64,
}
// RtOption (the tuning knobs 4..14 never change results and are left to the C header)
pub const RT_OPT_COUNT_TRAVERSAL: u32 = 1;
pub const RT_OPT_TIME_KERNELS: u32 = 2;
pub const RT_OPT_SAMPLES_IN_FLIGHT: u32 = 3;
pub const RT_OPT_FAST_SLAB: u32 = 15; // not bit-exact when set
pub const RT_LEAF_BIT: u32 = 0x8000_0000;
pub const RT_NO_INDEX: u32 = 0xFFFF_FFFF;
pub const RT_RNG_SEED: u64 = 0x0ef6_f79e_d30b_a75a; // src/rng.rs:10

unsafe extern "C" {
    pub fn rt_create(device_ordinal: c_int, out: *mut *mut RtContext) -> c_int;
    pub fn rt_destroy(ctx: *mut RtContext);
    pub fn rt_last_error(ctx: *const RtContext) -> *const c_char;
    pub fn rt_upload_scene(ctx: *mut RtContext, scene: *const RtSceneDesc) -> c_int;
    pub fn rt_render(
        ctx: *mut RtContext, params: *const RtRenderParams, rgb8: *mut u8,
        progress: Option<unsafe extern "C" fn(u32, *mut c_void)>, user: *mut c_void,
    ) -> c_int;
    /// Optional: the per-pixel fold of src/lib.rs:274-279 carried on from `sums` (samples
    /// `0..sample_begin`) over `sample_count` more samples; with `rt_resolve` a progressive render
    /// that does not average stored bytes.  `sums`: 3 f32 per pixel, kept by the caller.
    pub fn rt_render_sums_onto(
        ctx: *mut RtContext, params: *const RtRenderParams, sample_begin: u32, sample_count: u32,
        sums: *mut f32, sums_on_device: c_int,
    ) -> c_int;
    /// The two halves of `rt_render_sums_onto`: trace now (as many samples as fit; `*traced`),
    /// fold later, onto sums that another GPU has carried up to this GPU's first sample.
    pub fn rt_trace_samples(
        ctx: *mut RtContext, params: *const RtRenderParams, sample_begin: u32, sample_count: u32,
        traced: *mut u32,
    ) -> c_int;
    pub fn rt_reserve_samples(
        ctx: *mut RtContext, params: *const RtRenderParams, sample_count: u32, in_flight: *mut u32,
    ) -> c_int;
    pub fn rt_fold_samples(
        ctx: *mut RtContext, params: *const RtRenderParams, carry_on: c_int, sums: *mut f32,
        sums_on_device: c_int,
    ) -> c_int;
    pub fn rt_resolve(
        ctx: *mut RtContext, params: *const RtRenderParams, sums: *const f32, sums_on_device: c_int,
        n_samples: u32, running_average: c_int, sample_iteration: u32, rgb8: *mut u8,
    ) -> c_int;
    /// Optional: `BvhNode::new` (src/bvh.rs:48) for the triangle slice on the GPU.  Permutes
    /// `shapes` in place like the host builder; `*nodes` is malloc'd, release it with `rt_free`.
    pub fn rt_build_bvh_triangles(
        ctx: *mut RtContext, shapes: *mut RtTriangle, n: u32, nodes: *mut *mut RtNode, n_nodes: *mut u32,
    ) -> c_int;
    pub fn rt_build_bvh_spheres(
        ctx: *mut RtContext, shapes: *mut RtSphere, n: u32, nodes: *mut *mut RtNode, n_nodes: *mut u32,
    ) -> c_int;
    pub fn rt_free(p: *mut c_void);
    /// Page-locked pixel buffers (full-rate copies to and from the GPU); release with `rt_host_free`.
    pub fn rt_host_alloc(bytes: usize) -> *mut c_void;
    pub fn rt_host_free(p: *mut c_void);
    pub fn rt_abi_version() -> u32;
    pub fn rt_set_option(ctx: *mut RtContext, option: u32, value: u64) -> c_int;
    pub fn rt_render_sums(
        ctx: *mut RtContext, params: *const RtRenderParams, sample_begin: u32, sample_count: u32,
        sums: *mut f32, sums_on_device: c_int,
    ) -> c_int;
    pub fn rt_trace_primary(
        ctx: *mut RtContext, params: *const RtRenderParams, sample_index: u32, jitter: c_int, out: *mut RtHitRecord,
    ) -> c_int;
    pub fn rt_trace_rays(
        ctx: *mut RtContext, origins: *const f32, directions: *const f32, n_rays: u64, out: *mut RtHitRecord,
    ) -> c_int;
    pub fn rt_trace_rays_noalloc(
        ctx: *mut RtContext, origins: *const f32, directions: *const f32, n_rays: u64, out: *mut RtHitRecord,
        on_device: c_int,
    ) -> c_int;
    pub fn rt_self_test(ctx: *mut RtContext, which: u32, n_cases: u64, failures: *mut u64) -> c_int;
    pub fn rt_get_stats(ctx: *mut RtContext, out: *mut RtStats) -> c_int;
    pub fn rt_reset_stats(ctx: *mut RtContext) -> c_int;
    // Several GPUs behind one handle: `Scene::render` over a list of GPUs, same bytes as one GPU.
    pub fn rt_group_create(device_ordinals: *const c_int, n_devices: u32, out: *mut *mut RtGroup) -> c_int;
    pub fn rt_group_destroy(group: *mut RtGroup);
    pub fn rt_group_last_error(group: *const RtGroup) -> *const c_char;
    pub fn rt_group_size(group: *const RtGroup) -> u32;
    pub fn rt_group_member(group: *mut RtGroup, index: u32) -> *mut RtContext;
    pub fn rt_group_set_option(group: *mut RtGroup, option: u32, value: u64) -> c_int;
    pub fn rt_group_upload_scene(group: *mut RtGroup, scene: *const RtSceneDesc) -> c_int;
    pub fn rt_group_render(
        group: *mut RtGroup, params: *const RtRenderParams, rgb8: *mut u8,
        progress: Option<unsafe extern "C" fn(u32, *mut c_void)>, user: *mut c_void,
    ) -> c_int;
    pub fn rt_group_render_sums(
        group: *mut RtGroup, params: *const RtRenderParams, sample_begin: u32, sample_count: u32, sums: *mut f32,
    ) -> c_int;
    pub fn rt_group_get_stats(group: *mut RtGroup, out: *mut RtStats) -> c_int;
    pub fn rt_group_reset_stats(group: *mut RtGroup) -> c_int;
}

/// What `Scene::render` (src/lib.rs:188) becomes: flatten, upload, render into
/// the mmap'd pixel slice.  `flatten_*` are straight field copies:
///   BvhNode { kind: Branch { children }, min, max } -> RtNode { min, a: children[0], max, b: children[1] }
///   BvhNode { kind: Leaf { shapes_range }, .. }     -> RtNode { a: start, b: (end - start) | RT_LEAF_BIT }
///   Triangle { a, e1, e2, normals_texture_coordinates, material_index } -> RtTriangle with
///     RT_NO_INDEX for the indices its NormalsTextureCoordinates variant lacks.
pub unsafe fn render_on_gpu(desc: &RtSceneDesc, params: &RtRenderParams, pixels: &mut [u8]) {
    let mut ctx = std::ptr::null_mut();
    assert_eq!(unsafe { rt_create(0, &mut ctx) }, 0, "no usable GPU");
    let check = |rc: c_int| {
        assert!(rc == 0, "librt_mi355: {}", unsafe { CStr::from_ptr(rt_last_error(ctx)) }.to_string_lossy());
    };
    check(unsafe { rt_upload_scene(ctx, desc) });
    unsafe extern "C" fn progress(samples: u32, _: *mut c_void) {
        print!("\rSamples: {samples}"); // src/lib.rs:238-246
    }
    check(unsafe { rt_render(ctx, params, pixels.as_mut_ptr(), Some(progress), std::ptr::null_mut()) });
    unsafe { rt_destroy(ctx) };
}
