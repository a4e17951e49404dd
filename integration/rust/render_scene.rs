// Shipped as SOURCE ONLY: the build image has no Rust toolchain, so this file has never been compiled here.
// It is the binding INTEGRATION.md describes; the same C ABI (include/rbrt_hip.h) is exercised by the C++ host
// (rbrt_amd/host/render.cpp) and by the ctypes mirror (rbrt_amd/abi.py, checked against the header's layout).
//
// Replacement body of `render_scene` in rbrt_lib/src/lib.rs (reference lib.rs:75-124). It reads the scene exactly
// as the reference stores it (scene.rs:12-16): spheres and BasicTriangles come out of `scene.elements` through
// `Intersectable::as_sphere` / `as_basic_triangle` (trait_additions.rs), meshes out of `scene.triangle_meshes`.

pub fn render_scene(cam: Camera, num_samples: u32, scene: Scene) -> image::ImageBuffer<Rgb<u8>, Vec<u8>> {
    use crate::hip_ffi::*;
    println!("Starting rendering...");
    let v3 = |v: Vec3| [v.x, v.y, v.z];
    // Scene::hit tests `elements` in order, then `triangle_meshes` in order (scene.rs:23-41): keep both orders.
    // `elements` holds Spheres and (never from the YAML factory, but the type admits them) BasicTriangles: both go over,
    // with the order they are tested in (an earlier element keeps a tie in distance).
    let mut spheres: Vec<RbrtSphere> = Vec::new();
    let mut triangles: Vec<RbrtTriangle> = Vec::new();
    let mut order: Vec<u32> = Vec::new();
    for e in scene.elements.iter() {
        if let Some(s) = e.as_sphere() {
            order.push(spheres.len() as u32);
            spheres.push(RbrtSphere { center: v3(s.center), radius: s.radius, mat: s.material.as_ffi() });
        } else if let Some(t) = e.as_basic_triangle() {
            order.push(0x8000_0000u32 | triangles.len() as u32);
            triangles.push(RbrtTriangle { corners: [v3(t.corners[0]), v3(t.corners[1]), v3(t.corners[2])], mat: t.material.as_ffi() });
        } else {
            panic!("the GPU path knows Sphere and BasicTriangle elements (the only Intersectable types in rbrt_lib)");
        }
    }
    let meshes: Vec<RbrtMesh> = scene.triangle_meshes.iter().map(|m| RbrtMesh {
        n_total: m.is_padding_triangle.len() as u32,
        n_real: m.is_padding_triangle.iter().filter(|p| !**p).count() as u32,
        v0x: m.vertices[0][0].as_ptr(), v0y: m.vertices[0][1].as_ptr(), v0z: m.vertices[0][2].as_ptr(),
        e1x: m.edges[0][0].as_ptr(),    e1y: m.edges[0][1].as_ptr(),    e1z: m.edges[0][2].as_ptr(),
        e2x: m.edges[1][0].as_ptr(),    e2y: m.edges[1][1].as_ptr(),    e2z: m.edges[1][2].as_ptr(),
        nx: m.normals[0].as_ptr(), ny: m.normals[1].as_ptr(), nz: m.normals[2].as_ptr(),
        is_padding: m.is_padding_triangle.as_ptr() as *const u8,   // Vec<bool>: one byte per element, 0 or 1
        bbox_lo: v3(m.bbox.lower_bound), bbox_hi: v3(m.bbox.upper_bound),
        mat: m.material.as_ffi(),
    }).collect();
    let ffi_scene = RbrtScene { n_spheres: spheres.len() as u32, spheres: spheres.as_ptr(),
                                n_meshes: meshes.len() as u32, meshes: meshes.as_ptr(),
                                n_triangles: triangles.len() as u32, triangles: triangles.as_ptr(),
                                element_order: order.as_ptr() };
    let ffi_cam = RbrtCamera { position: v3(cam.position), right: v3(cam.right), up: v3(cam.up),
        img_center_point: v3(cam.img_center_point), mm_per_pix_hor: cam.mm_per_pix_hor,
        mm_per_pix_vert: cam.mm_per_pix_vert, img_width_pix: cam.img_width_pix, img_height_pix: cam.img_height_pix };
    let mut opts = std::mem::MaybeUninit::<RbrtRenderOpts>::uninit();
    let mut rgb = vec![0u8; (cam.img_width_pix * cam.img_height_pix * 3) as usize];
    let rc = unsafe {
        rbrt_render_opts_default(opts.as_mut_ptr());
        let mut opts = opts.assume_init();
        opts.spp = num_samples;                       // depth 50, 0.001/2000, bg (0.05,0.05,0.8) are the defaults
        rbrt_hip_render(&ffi_cam, &ffi_scene, &opts, std::ptr::null_mut(), rgb.as_mut_ptr())
    };
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(rbrt_hip_last_error()) }.to_string_lossy().into_owned();
        panic!("rbrt_hip_render failed ({}): {}", rc, msg);   // -6 = the reference's own "Encountered NAN" panic
    }
    println!("\rRendering 100% complete!");
    image::ImageBuffer::from_raw(cam.img_width_pix, cam.img_height_pix, rgb).unwrap()   // row-major RGB8, row 0 on top
}
