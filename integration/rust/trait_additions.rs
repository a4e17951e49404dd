// Shipped as SOURCE ONLY (no Rust toolchain in the build image: never compiled here).
//
// The two trait methods the binding needs, as additions to the reference's own files. The reference keeps its scene
// as trait objects -- `Scene { elements: Vec<Box<dyn Intersectable + Sync>>, triangle_meshes: Vec<TriangleMesh>,
// lights: Vec<Light> }` (rbrt_lib/src/scene.rs:12-16), materials as `Box<dyn RayScattering + Sync>`
// (sphere.rs:9, mesh.rs:24) -- so the concrete parameters have to be asked back through the traits.

// ---- rbrt_lib/src/lib.rs:38-41, trait Intersectable: one defaulted method --------------------------------------
pub trait Intersectable: Sync {
    fn intersect_with_ray(&self, ray: &Ray, min_dist: f32, max_dist: f32) -> Option<HitInformation>;
    /// The GPU path takes the concrete parameters of the two Intersectable types rbrt_lib has (the scene builder,
    /// blueprints.rs:144-149, only ever creates spheres; a BasicTriangle reaches `elements` only from code).
    fn as_sphere(&self) -> Option<&crate::sphere::Sphere> { None }
    fn as_basic_triangle(&self) -> Option<&crate::triangle::BasicTriangle> { None }
}

// ---- rbrt_lib/src/sphere.rs:12, inside `impl Intersectable for Sphere` ---------------------------------------------
//     fn as_sphere(&self) -> Option<&Sphere> { Some(self) }
// ---- rbrt_lib/src/triangle.rs:412, inside `impl Intersectable for BasicTriangle` ---------------------------------
//     fn as_basic_triangle(&self) -> Option<&BasicTriangle> { Some(self) }

// ---- rbrt_lib/src/materials.rs:4-12, trait RayScattering: one required method ------------------------------------
pub trait RayScattering {
    fn scatter(&self, incoming_ray: &Ray, hit_info: &HitInformation, attentuation: &mut Vec3, scattered_ray: &mut Ray) -> bool;
    fn as_ffi(&self) -> crate::hip_ffi::RbrtMaterial;
}
// lambertian.rs:10  fn as_ffi(&self) -> RbrtMaterial { RbrtMaterial { kind: 0, albedo: [self.albedo.x, self.albedo.y, self.albedo.z], param: 0.0 } }
// metal.rs:11       fn as_ffi(&self) -> RbrtMaterial { RbrtMaterial { kind: 1, albedo: [self.albedo.x, self.albedo.y, self.albedo.z], param: self.roughness } }
// dielectric.rs:10  fn as_ffi(&self) -> RbrtMaterial { RbrtMaterial { kind: 2, albedo: [0.0; 3], param: self.ref_idx } }
