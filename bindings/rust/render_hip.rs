//! `src/hip.rs` for kevidgel/pathtracer: `Camera::render` on an MI355X through `ptcore-sys`.
//!
//! This is the module a maintainer of the reference adds next to `camera.rs`.  It is written against the
//! reference's own types and has not been compiled (no Rust toolchain where the library is built); what it
//! has to produce is pinned down by the C++ walk of the same structures in
//! `pathtracer_amd/host/pthost.cpp` (`FlatScene::build`), which the tests run.
//!
//! What the reference has to expose for it (all of it `pub(crate)`, nothing changes behaviour):
//!   * `Triangle::data()`, `Triangle::material()`            (src/objects/tri_mesh.rs:22-26)
//!   * `Sphere::{center, radius, mat}`                        (src/objects/sphere.rs:11-16)
//!   * `Quad::{origin, u, v, normal, w, d, area, mat}`        (src/objects/quad_mesh.rs:12-22)
//!   * `Instance::{obj, transform, inverse}`                  (src/objects.rs:282-286)
//!   * `BBox::{min, max}`                                     (src/bvh.rs:34-37)
//!   * the private fields of `Camera`                         (src/camera.rs:34-61; same module, no change)
//!   * `Material::describe(&self, textures: &mut Textures) -> pt_material_desc` and
//!     `Texture::describe(&self, textures: &mut Textures) -> pts_texture`, one line per implementation
//!     (type tag + constructor arguments + the constants the constructors precompute), because
//!     `Arc<dyn Material>` and `Arc<dyn Texture>` are opaque (src/materials.rs:80, src/textures.rs:10).

use std::collections::HashMap;
use std::ffi::CStr;
use std::sync::{Arc, Mutex};

use image::RgbImage;
use ptcore_sys::*;

use crate::bvh::FlatBVHNode;
use crate::camera::Camera;
use crate::materials::MaterialRef;
use crate::objects::{InnerPrimitiveBuffer, LightBuffer, PrimitiveBuffer};
use crate::textures::TextureRef;

/// Textures by identity of their `Arc`, in first-use order; nested textures (Checkered) recurse through it.
#[derive(Default)]
pub struct Textures {
    index: HashMap<*const (), i32>,
    pub records: Vec<pts_texture>,
    pub images: Vec<pts_image>,
    pub texels: Vec<u8>,
}

impl Textures {
    pub fn id(&mut self, t: &TextureRef) -> i32 {
        let key = Arc::as_ptr(t) as *const ();
        if let Some(&i) = self.index.get(&key) {
            return i;
        }
        let rec = t.describe(self); // may add the textures it refers to first
        self.records.push(rec);
        let i = (self.records.len() - 1) as i32;
        self.index.insert(key, i);
        i
    }
    /// `Image`'s texels, RGB8 rows top to bottom (src/textures/image.rs:31-65 reads them that way).
    pub fn add_image(&mut self, img: &RgbImage) -> i32 {
        self.images.push(pts_image { width: img.width(), height: img.height(), offset: self.texels.len() as u64 });
        self.texels.extend_from_slice(img.as_raw());
        (self.images.len() - 1) as i32
    }
}

/// Everything `pt_scene_desc` points at.  Keep it alive until `pt_upload_scene` has returned.
#[derive(Default)]
pub struct FlatScene {
    tri_data: Vec<f32>, // [n][8][3], Triangle.data row by row
    tri_mat: Vec<i32>,
    spheres: Vec<f32>, // [n][4]
    sphere_mat: Vec<i32>,
    quads: Vec<f32>, // [n][17]
    quad_mat: Vec<i32>,
    instances: Vec<pt_instance_desc>,
    nodes: Vec<pt_node>,
    sets: Vec<pt_primset_desc>,
    materials: Vec<pt_material_desc>,
    material_index: HashMap<*const (), i32>,
    textures: Textures,
}

impl FlatScene {
    /// `objects` must have had `build_bvh()` called (src/main.rs:72): the primitives are copied in the
    /// order the builder left them, which is what the node arrays index.
    pub fn describe(objects: &PrimitiveBuffer, lights: &LightBuffer) -> FlatScene {
        let mut s = FlatScene::default();
        s.sets.push(pt_primset_desc::default()); // [0] objects
        s.sets.push(pt_primset_desc::default()); // [1] lights
        let top = s.add_set(objects, true);
        s.sets[0] = top;
        let l = s.add_set(&lights.objects, false);
        s.sets[1] = l;
        s
    }

    fn material(&mut self, m: &Option<MaterialRef>) -> i32 {
        let Some(m) = m else { return -1 };
        let key = Arc::as_ptr(m) as *const ();
        if let Some(&i) = self.material_index.get(&key) {
            return i;
        }
        let rec = m.describe(&mut self.textures);
        self.materials.push(rec);
        let i = (self.materials.len() - 1) as i32;
        self.material_index.insert(key, i);
        i
    }

    fn nodes_of<T: crate::objects::Hittable>(&mut self, b: &InnerPrimitiveBuffer<T>, first: usize) -> pt_buffer_desc {
        let mut d = pt_buffer_desc { prim_first: first as u32, prim_count: b.len() as u32, node_first: self.nodes.len() as u32, node_count: 0 };
        if let Some(bvh) = &b.bvh {
            for n in bvh.iter() {
                self.nodes.push(match n {
                    FlatBVHNode::Interior { right_child_offset, axis, bbox } => pt_node {
                        bmin: bbox.min.coords.into(), bmax: bbox.max.coords.into(), a: *right_child_offset as u32, b: *axis as u32,
                    },
                    FlatBVHNode::Leaf { offset, size, bbox } => pt_node {
                        bmin: bbox.min.coords.into(), bmax: bbox.max.coords.into(), a: *offset as u32, b: PT_NODE_LEAF | *size as u32,
                    },
                });
            }
            d.node_count = bvh.len() as u32;
        }
        d
    }

    /// One PrimitiveBuffer; `may_hold_instances` is false for the children of instances and for lights
    /// (the library takes one level of instancing, which is what the reference's scenes use; deeper nesting would
    /// have to be lifted here, transforms composed, as libpthost does; lights that are instances have pdf 0).
    fn add_set(&mut self, p: &PrimitiveBuffer, may_hold_instances: bool) -> pt_primset_desc {
        let mut set = pt_primset_desc::default();
        let first = self.tri_mat.len();
        for t in &p.triangles.buffer {
            let d = t.data(); // 8x3, row = attribute, column = vertex (src/objects/tri_mesh.rs:47-72)
            for row in 0..8 {
                for col in 0..3 {
                    self.tri_data.push(d[(row, col)]);
                }
            }
            let m = self.material(t.material());
            self.tri_mat.push(m);
        }
        set.tris = self.nodes_of(&p.triangles, first);
        let first = self.sphere_mat.len();
        for s in &p.spheres.buffer {
            self.spheres.extend_from_slice(&[s.center.x, s.center.y, s.center.z, s.radius]);
            let m = self.material(&s.mat);
            self.sphere_mat.push(m);
        }
        set.spheres = self.nodes_of(&p.spheres, first);
        let first = self.quad_mat.len();
        for q in &p.quads.buffer {
            for v in [&q.origin.coords, &q.u, &q.v, &q.normal, &q.w] {
                self.quads.extend_from_slice(&[v.x, v.y, v.z]);
            }
            self.quads.extend_from_slice(&[q.d, q.area]);
            let m = self.material(&q.mat);
            self.quad_mat.push(m);
        }
        set.quads = self.nodes_of(&p.quads, first);
        if may_hold_instances {
            let first = self.instances.len();
            // reserve the records first so that the instance BVH's offsets stay valid, then fill the children in
            for _ in &p.instances.buffer {
                self.instances.push(pt_instance_desc { transform: [0.0; 16], inverse: [0.0; 16], child_set: 0, chain_first: 0, chain_count: 0, reserved: 0 });
            }
            for (k, inst) in p.instances.buffer.iter().enumerate() {
                let child = self.add_set(&inst.obj, false);
                self.sets.push(child);
                let rec = &mut self.instances[first + k];
                // nalgebra stores column-major; the descriptor wants rows
                rec.transform.copy_from_slice(inst.transform.transpose().as_slice());
                rec.inverse.copy_from_slice(inst.inverse.transpose().as_slice());
                rec.child_set = (self.sets.len() - 1) as u32;
            }
            set.instances = self.nodes_of(&p.instances, first);
        }
        set
    }

    pub fn desc(&self) -> pt_scene_desc {
        pt_scene_desc {
            n_tris: self.tri_mat.len() as u32, tri_data: self.tri_data.as_ptr(), tri_mat: self.tri_mat.as_ptr(),
            n_spheres: self.sphere_mat.len() as u32, spheres: self.spheres.as_ptr(), sphere_mat: self.sphere_mat.as_ptr(),
            n_quads: self.quad_mat.len() as u32, quads: self.quads.as_ptr(), quad_mat: self.quad_mat.as_ptr(),
            n_instances: self.instances.len() as u32, instances: self.instances.as_ptr(),
            n_nodes: self.nodes.len() as u32, nodes: self.nodes.as_ptr(),
            n_sets: self.sets.len() as u32, sets: self.sets.as_ptr(),
            n_materials: self.materials.len() as u32, materials: self.materials.as_ptr(),
            n_textures: self.textures.records.len() as u32, textures: self.textures.records.as_ptr(),
            n_images: self.textures.images.len() as u32, images: self.textures.images.as_ptr(),
            n_texel_bytes: self.textures.texels.len() as u64, texels: self.textures.texels.as_ptr(),
            n_media: 0, media: std::ptr::null(), n_moving: 0, moving: std::ptr::null(),
        }
    }
}

/// The GPUs a render runs on (`pt_group`: one context per device, strata split between them, one RCCL
/// reduce of the accumulators when the image is read); destroyed when dropped; errors carry the library's text.
pub struct Devices(*mut pt_group);

impl Devices {
    pub fn new(devices: &[i32]) -> Result<Devices, String> {
        let mut g = std::ptr::null_mut();
        let rc = unsafe { pt_group_create(devices.as_ptr(), devices.len() as i32, &mut g) };
        if rc != PT_OK {
            return Err(last_error(std::ptr::null()));
        }
        Ok(Devices(g))
    }
    fn check(&self, rc: i32) -> Result<(), String> {
        if rc == PT_OK { Ok(()) } else { Err(last_error(self.0)) }
    }
}

impl Drop for Devices {
    fn drop(&mut self) {
        unsafe { pt_group_destroy(self.0) }
    }
}

fn last_error(g: *const pt_group) -> String {
    unsafe { CStr::from_ptr(pt_group_last_error(g)) }.to_string_lossy().into_owned()
}

impl Camera {
    fn describe(&self) -> pt_camera_desc {
        pt_camera_desc {
            image_width: self.image_width, image_height: self.image_height,
            center: self.center.coords.into(), pixel00: self.pixel00.coords.into(),
            pixel_du: self.pixel_du.into(), pixel_dv: self.pixel_dv.into(),
            defocus_disk_u: self.defocus_disk_u.into(), defocus_disk_v: self.defocus_disk_v.into(),
            defocus_angle: self.defocus_angle, sqrt_spp: self.sqrt_spp, inv_sqrt_spp: self.inv_sqrt_spp,
            max_depth: self.max_depth, background: self.background.into(),
        }
    }

    /// `Camera::render` (src/camera.rs:162-236) on the device: same arguments, same image.  The passes of
    /// the stratum loop (:176-198) are handed over a few hundred at a time and the shared image is refreshed
    /// in between (the reference refreshes it after every pass, :221-226).
    /// `devices`: the GPUs of this node to use, e.g. `&[0]` or `&[0, 1, 2, 3, 4, 5, 6, 7]`.
    pub fn render_hip(&self, objects: &PrimitiveBuffer, lights: &LightBuffer, buffer: Arc<Mutex<RgbImage>>, devices: &[i32], seed: u64) -> Result<(), String> {
        let flat = FlatScene::describe(objects, lights);
        let dev = Devices::new(devices)?;
        dev.check(unsafe { pt_group_upload_scene(dev.0, &flat.desc()) })?;
        dev.check(unsafe { pt_group_set_camera(dev.0, &self.describe()) })?;
        let total = self.sqrt_spp * self.sqrt_spp;
        let per_refresh = 256 * devices.len() as u32;
        let mut begin = 0;
        while begin < total {
            let end = (begin + per_refresh).min(total);
            dev.check(unsafe { pt_group_render(dev.0, begin, end, seed) })?;
            let mut img = buffer.lock().unwrap();
            dev.check(unsafe { pt_group_read_rgb8(dev.0, img.as_mut_ptr()) })?;
            begin = end;
        }
        Ok(())
    }
}
