// Shipped as SOURCE ONLY: the build image has no Rust toolchain, so this file has never been compiled here.
// It is the binding INTEGRATION.md describes; the same C ABI (include/rbrt_hip.h) is exercised by the C++ host
// (rbrt_amd/host/render.cpp) and by the ctypes mirror (rbrt_amd/abi.py, checked against the header's layout).
//
// Goes to rbrt_lib/src/hip_ffi.rs (add `pub mod hip_ffi;` to rbrt_lib/src/lib.rs).

use std::os::raw::{c_char, c_int};

#[repr(C)] #[derive(Copy, Clone)]
pub struct RbrtMaterial { pub kind: i32, pub albedo: [f32; 3], pub param: f32 }   // 0 lambertian, 1 metal, 2 dielectric
#[repr(C)] #[derive(Copy, Clone)]
pub struct RbrtSphere { pub center: [f32; 3], pub radius: f32, pub mat: RbrtMaterial }
#[repr(C)]
pub struct RbrtMesh {
    pub n_total: u32, pub n_real: u32,
    pub v0x: *const f32, pub v0y: *const f32, pub v0z: *const f32,      // vertices[0][0..3]
    pub e1x: *const f32, pub e1y: *const f32, pub e1z: *const f32,      // edges[0][0..3]
    pub e2x: *const f32, pub e2y: *const f32, pub e2z: *const f32,      // edges[1][0..3]
    pub nx: *const f32, pub ny: *const f32, pub nz: *const f32,         // normals[0..3]
    pub is_padding: *const u8,                                           // Vec<bool> is one byte per element
    pub bbox_lo: [f32; 3], pub bbox_hi: [f32; 3],
    pub mat: RbrtMaterial,
}
#[repr(C)] #[derive(Copy, Clone)]
pub struct RbrtTriangle { pub corners: [[f32; 3]; 3], pub mat: RbrtMaterial }       // BasicTriangle, triangle.rs:9-34
#[repr(C)]
pub struct RbrtScene {
    pub n_spheres: u32, pub spheres: *const RbrtSphere, pub n_meshes: u32, pub meshes: *const RbrtMesh,
    pub n_triangles: u32, pub triangles: *const RbrtTriangle,
    pub element_order: *const u32,   // null, or Scene::elements order: index, bit 31 set for a triangle (ABI version 2)
}
#[repr(C)]
pub struct RbrtCamera {
    pub position: [f32; 3], pub right: [f32; 3], pub up: [f32; 3], pub img_center_point: [f32; 3],
    pub mm_per_pix_hor: f32, pub mm_per_pix_vert: f32, pub img_width_pix: u32, pub img_height_pix: u32,
}
#[repr(C)]
pub struct RbrtRenderOpts {
    pub spp: u32, pub max_depth: u32, pub min_dist: f32, pub max_dist: f32, pub bg: [f32; 3],
    pub seed: u64, pub tile_rank: u32, pub tile_world: u32, pub flags: u32, pub reserved: u32,
}

#[link(name = "rbrt_hip")]
extern "C" {
    pub fn rbrt_render_opts_default(opts: *mut RbrtRenderOpts);
    pub fn rbrt_hip_render(cam: *const RbrtCamera, scene: *const RbrtScene, opts: *const RbrtRenderOpts,
                           out_radiance: *mut f32, out_rgb8: *mut u8) -> c_int;
    pub fn rbrt_hip_last_error() -> *const c_char;
    pub fn rbrt_hip_device_count() -> c_int;
}
