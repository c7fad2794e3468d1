//! `src/gpu_renderer.rs` for the reference crate: `GpuRenderer` has the surface of `Renderer`
//! (`src/renderer.rs:18-157`) and runs the per-pixel sampling loop on an MI355X through `rt-mi355x-sys`.
//!
//! Uncompiled (no Rust toolchain in the build image).  It needs these one-line accessors added to the
//! reference, because the fields are private to their modules:
//!   - `Camera::raw(&self) -> [[f32; 3]; 6]` + `defocus_angle()` (src/camera.rs:6-17)
//!   - `TriangleRef::{mesh(&self) -> &Arc<TriangleMeshData>, face(&self) -> usize}` and
//!     `TriangleMeshData::{vertices, face_indices, normals, uv, material}` as `pub(crate)`
//!     (src/object/triangle_mesh.rs:12-18,84-102)
//!   - `Sphere::{center, radius, material}` as `pub(crate)` (src/object/sphere.rs:11-18)
//!   - `Checkerboard::{inv_scale(&self) -> f32, even(&self) -> &Arc<Texture>, odd(&self) -> &Arc<Texture>}`
//!     (src/texture.rs:54-58) and `Image::image(&self) -> &DynamicImage` (src/texture.rs:85-87)
//! The material structs' fields (`Lambertian::texture`, `Metal::{texture, fuzz}`, `Dielectric::refraction_index`,
//! `DiffuseLight::{texture, strength}`, `Mix::{left, right, factor}`, `TrowbridgeReitz::roughness`) and
//! `SolidColor::albedo` are already `pub` (src/material.rs:43-46,67-70,85-87,124-127,179-184,207-209;
//! src/texture.rs:43-45).
//! and it must be handed the scene BEFORE `build_bvh` (the library builds the same tree itself,
//! src/bvh.rs:248-285), i.e. `main` skips `scene.build_bvh(...)` when this renderer is selected.
use std::collections::HashMap;
use std::ffi::CStr;
use std::ptr;
use std::sync::Arc;

use color_eyre::eyre::eyre;
use image::RgbImage;
use rt_mi355x_sys as sys;

use crate::camera::Camera;
use crate::material::Material;
use crate::object::Object;
use crate::renderer::{pixels_to_image, should_save_image, ImageOutput};
use crate::scene::{RenderSettings, SceneDescription};
use crate::texture::Texture;
use crate::Result;

fn check(code: i32) -> Result<()> {
    if code == sys::RT_OK {
        Ok(())
    } else {
        let msg = unsafe { CStr::from_ptr(sys::rt_last_error()) }.to_string_lossy().into_owned();
        Err(eyre!("rt_mi355x error {code}: {msg}"))
    }
}

/// Host-side copy of the object list in the layout `rt_scene_desc` points into.
#[derive(Default)]
struct Flattened {
    positions: Vec<Vec<f32>>,
    normals: Vec<Vec<f32>>,
    uvs: Vec<Vec<f32>>,
    indices: Vec<Vec<u32>>,
    mesh_material: Vec<i32>,
    spheres: Vec<sys::rt_sphere>,
    materials: Vec<sys::rt_material>,
    textures: Vec<sys::rt_texture>,
    texels: Vec<Vec<u8>>,
    material_ids: HashMap<*const Material, i32>,
    texture_ids: HashMap<*const Texture, i32>,
}

impl Flattened {
    fn texture(&mut self, t: &Arc<Texture>) -> i32 {
        if let Some(&id) = self.texture_ids.get(&Arc::as_ptr(t)) {
            return id;
        }
        let mut rec = sys::rt_texture {
            kind: sys::RT_TEX_SOLID, rgb: [0.0; 3], inv_scale: 0.0, even: -1, odd: -1, width: 0, height: 0, pixels: ptr::null(),
        };
        match &**t {
            Texture::SolidColor(c) => rec.rgb = c.albedo.to_array(),
            Texture::Checkerboard(c) => {
                rec.kind = sys::RT_TEX_CHECKER;
                rec.inv_scale = c.inv_scale();
                rec.even = self.texture(c.even());
                rec.odd = self.texture(c.odd());
            }
            Texture::Image(img) => {
                // channels 0..2 of get_pixel, as Image::value_at reads them (src/texture.rs:115-136)
                let rgb = img.image().to_rgb8();
                rec.kind = sys::RT_TEX_IMAGE;
                rec.width = rgb.width();
                rec.height = rgb.height();
                self.texels.push(rgb.into_raw());
                rec.pixels = self.texels.last().unwrap().as_ptr();
            }
        }
        self.textures.push(rec);
        let id = self.textures.len() as i32 - 1;
        self.texture_ids.insert(Arc::as_ptr(t), id);
        id
    }

    fn material(&mut self, m: &Arc<Material>) -> i32 {
        if let Some(&id) = self.material_ids.get(&Arc::as_ptr(m)) {
            return id;
        }
        let mut rec = sys::rt_material { kind: 0, texture: -1, param: 0.0, left: -1, right: -1 };
        match &**m {
            Material::Lambertian(l) => { rec.kind = sys::RT_MAT_LAMBERTIAN; rec.texture = self.texture(&l.texture); }
            Material::Metal(x) => { rec.kind = sys::RT_MAT_METAL; rec.texture = self.texture(&x.texture); rec.param = x.fuzz; }
            Material::Dielectric(d) => { rec.kind = sys::RT_MAT_DIELECTRIC; rec.param = d.refraction_index; }
            Material::DiffuseLight(d) => {
                rec.kind = sys::RT_MAT_DIFFUSE_LIGHT;
                rec.texture = self.texture(&d.texture);
                rec.param = d.strength;
            }
            Material::Mix(x) => {
                rec.kind = sys::RT_MAT_MIX;
                rec.param = x.factor;
                rec.left = self.material(&x.left);
                rec.right = self.material(&x.right);
            }
            Material::TrowbridgeReitz(t) => { rec.kind = sys::RT_MAT_TROWBRIDGE_REITZ; rec.param = t.roughness; }
        }
        self.materials.push(rec);
        let id = self.materials.len() as i32 - 1;
        self.material_ids.insert(Arc::as_ptr(m), id);
        id
    }

    /// The loader pushes every face of mesh 0, then mesh 1, ... (src/scene.rs:238-242): consecutive
    /// `TriangleRef`s of one `TriangleMeshData` become one `rt_mesh`, so primitive IDs equal list positions.
    fn push_objects(&mut self, objects: &[Object]) -> Result<()> {
        let mut current: *const () = ptr::null();
        for object in objects {
            match object {
                Object::TriangleRef(tri) => {
                    let data = tri.mesh();
                    let key = Arc::as_ptr(data) as *const ();
                    if key != current || tri.face() != self.indices.last().map_or(0, |i| i.len() / 3) {
                        current = key;
                        self.positions.push(data.vertices.iter().flat_map(|p| p.to_array()).collect());
                        self.normals.push(data.normals.iter().flat_map(|n| n.to_array()).collect());
                        self.uvs.push(data.uv.iter().flat_map(|t| [t.u, t.v]).collect());
                        self.indices.push(Vec::new());
                        let id = self.material(&data.material);
                        self.mesh_material.push(id);
                    }
                    let (a, b, c) = data.face_indices[tri.face()];
                    self.indices.last_mut().unwrap().extend([a, b, c]);
                }
                Object::Sphere(s) => {
                    let material = self.material(&s.material);
                    self.spheres.push(sys::rt_sphere { center: s.center.to_array(), radius: s.radius, material });
                }
                other => return Err(eyre!("GpuRenderer needs the flat object list, found {}", other.name())),
            }
        }
        Ok(())
    }
}

pub struct GpuRenderer {
    ctx: *mut sys::rt_context,
    scene: *mut sys::rt_scene,
    camera: sys::rt_camera,
    render: RenderSettings,
    seed: u64,
}

unsafe impl Send for GpuRenderer {}

impl GpuRenderer {
    /// `Renderer::new(camera, scene, render)` (src/renderer.rs:25-31); `scene.root_object` must still be the
    /// `World` the loader produced.
    pub fn new(camera: Camera, scene: SceneDescription, render: RenderSettings) -> Result<Self> {
        let Object::World(world) = &scene.root_object else {
            return Err(eyre!("GpuRenderer::new wants the scene before build_bvh"));
        };
        let mut flat = Flattened::default();
        flat.push_objects(&world.objects)?;
        let meshes: Vec<sys::rt_mesh> = (0..flat.indices.len())
            .map(|i| sys::rt_mesh {
                num_vertices: (flat.positions[i].len() / 3) as u32,
                positions: flat.positions[i].as_ptr(),
                num_normals: (flat.normals[i].len() / 3) as u32,
                normals: if flat.normals[i].is_empty() { ptr::null() } else { flat.normals[i].as_ptr() },
                num_uvs: (flat.uvs[i].len() / 2) as u32,
                uvs: if flat.uvs[i].is_empty() { ptr::null() } else { flat.uvs[i].as_ptr() },
                num_faces: (flat.indices[i].len() / 3) as u32,
                indices: flat.indices[i].as_ptr(),
                material: flat.mesh_material[i],
            })
            .collect();
        let desc = sys::rt_scene_desc {
            num_meshes: meshes.len() as u32,
            meshes: meshes.as_ptr(),
            num_spheres: flat.spheres.len() as u32,
            spheres: flat.spheres.as_ptr(),
            num_materials: flat.materials.len() as u32,
            materials: flat.materials.as_ptr(),
            num_textures: flat.textures.len() as u32,
            textures: flat.textures.as_ptr(),
            spheres_first: 0,
        };
        let mut ctx = ptr::null_mut();
        check(unsafe { sys::rt_context_create(0, ptr::null_mut(), 0, &mut ctx) })?;
        let mut handle = ptr::null_mut();
        if let Err(e) = check(unsafe { sys::rt_scene_create(ctx, &desc, &mut handle) }) {
            unsafe { sys::rt_context_destroy(ctx) };
            return Err(e);
        }
        let [center, pixel_00_loc, pixel_delta_u, pixel_delta_v, defocus_disk_u, defocus_disk_v] = camera.raw();
        let camera = sys::rt_camera {
            center, pixel_00_loc, pixel_delta_u, pixel_delta_v, defocus_disk_u, defocus_disk_v,
            defocus_angle: camera.defocus_angle(), z_near: camera.z_near, z_far: camera.z_far,
        };
        Ok(GpuRenderer { ctx, scene: handle, camera, render, seed: rand::random() })
    }

    fn settings(&self, mode: u32, begin: u32, count: u32) -> sys::rt_settings {
        sys::rt_settings {
            image_width: self.render.image_width,
            image_height: self.render.image_height,
            max_depth: self.render.max_depth,
            samples_per_pixel: self.render.samples_per_pixel,
            background_color: self.render.background_color.to_array(),
            mode,
            seed: self.seed,
            sample_begin: begin,
            sample_count: count,
        }
    }

    /// `Renderer::render(square_size)` (src/renderer.rs:72-103).
    pub fn render(&self, _square_size: usize) -> Result<RgbImage> {
        let (w, h) = (self.render.image_width, self.render.image_height);
        let mut sums = vec![0.0f32; (w * h * 3) as usize];
        let st = self.settings(sys::RT_MODE_LINEAR_MEAN, 0, self.render.samples_per_pixel);
        check(unsafe { sys::rt_render(self.ctx, self.scene, &self.camera, &st, sums.as_mut_ptr(), ptr::null_mut()) })?;
        let sample_scale = 1.0 / self.render.samples_per_pixel as f32;
        Ok(pixels_to_image(sums.into_iter().map(|c| c * sample_scale).collect(), w, h))
    }

    /// `Renderer::render_progressive(output, square_size)` (src/renderer.rs:105-156): one device call per
    /// report interval instead of one frame per sample.  The running `aggregate_image` (src/renderer.rs:110) lives
    /// on the GPU: `rt_render_device` adds every sample of the interval to it one by one in ascending sample order,
    /// the order of `aggregate_image[idx] += ...` (src/renderer.rs:135-137), so the sums do not depend on where the
    /// report intervals fall; each report downloads a copy.
    pub fn render_progressive(&self, mut output: ImageOutput, _square_size: usize) -> Result<()> {
        let (w, h, spp) = (self.render.image_width, self.render.image_height, self.render.samples_per_pixel);
        output.init(w, h)?;
        let len = (w * h * 3) as usize;
        let mut d_aggregate: *mut f32 = ptr::null_mut();
        check(unsafe { sys::rt_accum_create(self.ctx, w, h, &mut d_aggregate) })?;
        let mut aggregate = vec![0.0f32; len];
        let mut run = || -> Result<()> {
            let mut done = 0;
            while done < spp {
                let next = (done + 1..=spp).find(|s| should_save_image(*s, spp)).unwrap();
                let st = self.settings(sys::RT_MODE_GAMMA_PER_SAMPLE, done, next - done);
                check(unsafe {
                    sys::rt_render_device(self.ctx, self.scene, &self.camera, &st, d_aggregate, ptr::null_mut())
                })?;
                check(unsafe { sys::rt_accum_download(self.ctx, d_aggregate, len, aggregate.as_mut_ptr()) })?;
                done = next;
                let scale = 1.0 / done as f32;
                output.write(aggregate.iter().map(|c| c * scale).collect(), w, h)?;
            }
            Ok(())
        };
        let result = run();
        unsafe { sys::rt_accum_destroy(self.ctx, d_aggregate) };
        result
    }
}

impl Drop for GpuRenderer {
    fn drop(&mut self) {
        unsafe {
            sys::rt_scene_destroy(self.scene);
            sys::rt_context_destroy(self.ctx);
        }
    }
}
