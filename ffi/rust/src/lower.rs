//! Lowering of the reference's object tree to the flat arrays of `mirt_scene_desc`.
//!
//! Source only, never compiled (README.md).  It is written as a module of the reference crate: it reads private
//! fields (`Dielectric.index_of_refraction` src/material.rs:95, `CheckerTexture.{odd,even}` src/texture.rs:31-34,
//! `NoiseTexture.{noise,scale}` :62-65, `ImageTexture.{data,width,height}` :84-88, `Perlin.*` src/perlin.rs:5-10,
//! `World.data` src/world.rs:20-24, `BVH.{world,nodes}` src/bvh.rs:18-21).  The tested C++ twin of every item is the
//! item of the same name in raytracing_amd/host/rt.hpp.
use crate::ffi::*;
use crate::*; // Sphere, MovingSphere, XYRect, ..., World, BVH, Lambertian, ..., SolidTexture, ..., Vec3

#[derive(Default)]
pub(crate) struct Lowerer {
    pub spheres: Vec<mirt_sphere>,
    pub moving_spheres: Vec<mirt_moving_sphere>,
    pub rects: Vec<mirt_rect>,
    pub boxes: Vec<mirt_box>,
    pub chains: Vec<mirt_chain>, // chains[0] = the empty chain
    pub chain_ops: Vec<mirt_chain_op>,
    pub media: Vec<mirt_medium>,
    pub materials: Vec<mirt_material>,
    pub textures: Vec<mirt_texture>,
    pub perlins: Vec<mirt_perlin>,
    pub images: Vec<mirt_image>,
    pub lights: Vec<mirt_light>,
    pub light_groups: Vec<mirt_light_group>,
    pub lights_mode: u32,
    pub error: Option<String>, // the first construct the device path cannot express
    // state of the walk
    wrappers: Vec<mirt_chain_op>, // outermost first
    pub type_index: u32,
    pub object_index: u32,
    visit_counter: u32,
    boundary_mode: bool,
    last_boundary: Option<(u32, u32)>, // (tag, index)
    boundary_prims: u32,
}

fn v3(v: &Vec3) -> [f32; 3] { [v.x, v.y, v.z] }

impl Lowerer {
    pub fn new() -> Self { let mut l = Self::default(); l.chains.push(mirt_chain::default()); l }
    fn unsupported(&mut self, what: &str) { if self.error.is_none() { self.error = Some(what.to_string()); } }
    /// Equal chains / materials / textures are stored once (a linear search here; rt.hpp hashes the records'
    /// bytes, which a scene of millions of differently coloured spheres needs).
    fn current_chain(&mut self) -> u32 {
        if self.wrappers.is_empty() { return 0; }
        if self.wrappers.len() > MIRT_MAX_CHAIN_OPS { self.unsupported("more than MIRT_MAX_CHAIN_OPS nested wrappers"); return 0; }
        for (i, c) in self.chains.iter().enumerate() {
            let ops = &self.chain_ops[c.first_op as usize..(c.first_op + c.n_ops) as usize];
            if c.n_ops as usize == self.wrappers.len() && ops == &self.wrappers[..] { return i as u32; }
        }
        self.chains.push(mirt_chain { first_op: self.chain_ops.len() as u32, n_ops: self.wrappers.len() as u32 });
        self.chain_ops.extend_from_slice(&self.wrappers);
        self.chains.len() as u32 - 1
    }
    pub fn add_texture(&mut self, t: mirt_texture) -> u32 {
        if let Some(i) = self.textures.iter().position(|x| *x == t) { return i as u32; }
        self.textures.push(t);
        self.textures.len() as u32 - 1
    }
    pub fn add_material(&mut self, m: mirt_material) -> u32 {
        if let Some(i) = self.materials.iter().position(|x| *x == m) { return i as u32; }
        self.materials.push(m);
        self.materials.len() as u32 - 1
    }
    /// `order` = the place of the primitive in the container's visiting order: of two hits at the same t the later
    /// one wins, in World::hit (src/world.rs:75-94) as in the tournament of BVH::check_node (src/bvh.rs:109-131).
    fn header(&mut self, material: u32) -> mirt_prim_header {
        let h = mirt_prim_header { material, chain: self.current_chain(), order: self.visit_counter,
                                   type_index: self.type_index, object_index: self.object_index,
                                   flags: if self.boundary_mode { MIRT_PRIM_FLAG_BOUNDARY } else { 0 } };
        self.visit_counter += 1;
        h
    }
    fn note_boundary(&mut self, tag: u32, index: u32) {
        if self.boundary_mode { self.last_boundary = Some((tag, index)); self.boundary_prims += 1; }
    }
    /// The description the arrays add up to.  The Lowerer must outlive the upload call.
    pub fn desc(&self, time0: f32, time1: f32) -> mirt_scene_desc {
        mirt_scene_desc {
            spheres: self.spheres.as_ptr(), n_spheres: self.spheres.len() as u32,
            moving_spheres: self.moving_spheres.as_ptr(), n_moving_spheres: self.moving_spheres.len() as u32,
            rects: self.rects.as_ptr(), n_rects: self.rects.len() as u32,
            boxes: self.boxes.as_ptr(), n_boxes: self.boxes.len() as u32,
            chains: self.chains.as_ptr(), n_chains: self.chains.len() as u32,
            chain_ops: self.chain_ops.as_ptr(), n_chain_ops: self.chain_ops.len() as u32,
            media: self.media.as_ptr(), n_media: self.media.len() as u32,
            materials: self.materials.as_ptr(), n_materials: self.materials.len() as u32,
            textures: self.textures.as_ptr(), n_textures: self.textures.len() as u32,
            perlins: self.perlins.as_ptr(), n_perlins: self.perlins.len() as u32,
            images: self.images.as_ptr(), n_images: self.images.len() as u32,
            lights_mode: self.lights_mode,
            lights: self.lights.as_ptr(), n_lights: self.lights.len() as u32,
            light_groups: self.light_groups.as_ptr(), n_light_groups: self.light_groups.len() as u32,
            time0, time1, build_flags: MIRT_BUILD_AUTO,
        }
    }
}

// ---- textures (src/texture.rs) and materials (src/material.rs) --------------------------------------------------
pub(crate) trait LowerTexture { fn lower(&self, l: &mut Lowerer) -> u32; }
pub(crate) trait LowerMaterial { fn lower(&self, l: &mut Lowerer) -> u32; }

impl LowerTexture for SolidTexture {
    fn lower(&self, l: &mut Lowerer) -> u32 {
        l.add_texture(mirt_texture { kind: MIRT_TEX_SOLID, rgb: v3(&self.color), ..Default::default() })
    }
}
impl<O: Texture + LowerTexture, E: Texture + LowerTexture> LowerTexture for CheckerTexture<O, E> {
    fn lower(&self, l: &mut Lowerer) -> u32 {
        let (odd, even) = (self.odd.lower(l), self.even.lower(l));
        l.add_texture(mirt_texture { kind: MIRT_TEX_CHECKER, odd, even, ..Default::default() })
    }
}
impl LowerTexture for NoiseTexture {
    fn lower(&self, l: &mut Lowerer) -> u32 {
        let p = &self.noise; // src/perlin.rs:5-10
        let mut t = mirt_perlin { random_vec: [[0.0; 3]; 256], perm_x: [0; 256], perm_y: [0; 256], perm_z: [0; 256] };
        for i in 0..256 {
            t.random_vec[i] = v3(&p.random_vec[i]);
            t.perm_x[i] = p.perm_x[i]; t.perm_y[i] = p.perm_y[i]; t.perm_z[i] = p.perm_z[i];
        }
        l.perlins.push(t);
        l.add_texture(mirt_texture { kind: MIRT_TEX_NOISE, scale: self.scale, index: l.perlins.len() as u32 - 1, ..Default::default() })
    }
}
impl LowerTexture for ImageTexture {
    fn lower(&self, l: &mut Lowerer) -> u32 {
        // `data` is what image::open(..).as_bytes() gave: RGB8, row 0 = top (src/texture.rs:91-98); it stays alive in the
        // scene object for the duration of the upload
        l.images.push(mirt_image { rgb8: self.data.as_ptr(), width: self.width, height: self.height });
        l.add_texture(mirt_texture { kind: MIRT_TEX_IMAGE, index: l.images.len() as u32 - 1, ..Default::default() })
    }
}
impl<T: Texture + LowerTexture> LowerMaterial for Lambertian<T> {
    fn lower(&self, l: &mut Lowerer) -> u32 { let t = self.albedo.lower(l); l.add_material(mirt_material { kind: MIRT_MAT_LAMBERTIAN, texture: t, param: 0.0 }) }
}
impl<T: Texture + LowerTexture> LowerMaterial for Metal<T> {
    // `fuzz` is already clamped to <= 1 by Metal::new (src/material.rs:67-72)
    fn lower(&self, l: &mut Lowerer) -> u32 { let t = self.albedo.lower(l); l.add_material(mirt_material { kind: MIRT_MAT_METAL, texture: t, param: self.fuzz }) }
}
impl LowerMaterial for Dielectric {
    fn lower(&self, l: &mut Lowerer) -> u32 { l.add_material(mirt_material { kind: MIRT_MAT_DIELECTRIC, texture: 0, param: self.index_of_refraction }) }
}
impl<T: Texture + LowerTexture> LowerMaterial for DiffuseLight<T> {
    fn lower(&self, l: &mut Lowerer) -> u32 { let t = self.emit.lower(l); l.add_material(mirt_material { kind: MIRT_MAT_DIFFUSE_LIGHT, texture: t, param: 0.0 }) }
}
impl<T: Texture + LowerTexture> LowerMaterial for Isotropic<T> {
    // no scatter() in the reference (src/material.rs:172-180): the device absorbs too
    fn lower(&self, l: &mut Lowerer) -> u32 { let t = self.albedo.lower(l); l.add_material(mirt_material { kind: MIRT_MAT_ISOTROPIC, texture: t, param: 0.0 }) }
}

// ---- objects (src/objects.rs), wrappers (src/transforms.rs), containers (src/world.rs, src/bvh.rs) ----------------
pub(crate) trait Lower {
    fn lower(&self, l: &mut Lowerer);
    /// As a member of the `lights` argument: what Hittable::pdf_value / random do for it (src/hittable.rs:52-57).
    fn lower_light(&self, l: &mut Lowerer) { l.lights.push(mirt_light { kind: MIRT_LIGHT_OTHER, p: [0.0; 5] }); }
    /// As the `lights` argument itself (src/renderer.rs:47): a bare object.
    fn lower_lights_argument(&self, l: &mut Lowerer) { l.lights_mode = MIRT_LIGHTS_OBJECT; self.lower_light(l); }
}

impl<M: Material + LowerMaterial> Lower for Sphere<M> {
    fn lower(&self, l: &mut Lowerer) {
        let m = self.material.lower(l);
        let h = l.header(m);
        l.spheres.push(mirt_sphere { h, center: v3(&self.center), radius: self.radius });
        l.note_boundary(MIRT_PRIM_SPHERE, l.spheres.len() as u32 - 1);
    }
    fn lower_light(&self, l: &mut Lowerer) { // src/objects.rs:82-100
        l.lights.push(mirt_light { kind: MIRT_LIGHT_SPHERE, p: [self.center.x, self.center.y, self.center.z, self.radius, 0.0] });
    }
}
impl<M: Material + LowerMaterial> Lower for MovingSphere<M> {
    fn lower(&self, l: &mut Lowerer) {
        let m = self.material.lower(l);
        let h = l.header(m);
        l.moving_spheres.push(mirt_moving_sphere { h, center0: v3(&self.center0), center1: v3(&self.center1),
                                                   time0: self.time0, time1: self.time1, radius: self.radius });
        l.note_boundary(MIRT_PRIM_MOVING_SPHERE, l.moving_spheres.len() as u32 - 1);
    }
}
macro_rules! lower_rect {
    ($ty:ident, $axis:expr, $a0:ident, $a1:ident, $b0:ident, $b1:ident) => {
        impl<M: Material + LowerMaterial> $ty<M> {
            fn push_rect(&self, l: &mut Lowerer) {
                let m = self.material.lower(l);
                let h = l.header(m);
                l.rects.push(mirt_rect { h, axis: $axis, a0: self.$a0, a1: self.$a1, b0: self.$b0, b1: self.$b1, k: self.k });
                l.note_boundary(MIRT_PRIM_RECT, l.rects.len() as u32 - 1);
            }
        }
    };
}
lower_rect!(XYRect, MIRT_RECT_XY, x0, x1, y0, y1);
lower_rect!(XZRect, MIRT_RECT_XZ, x0, x1, z0, z1);
lower_rect!(YZRect, MIRT_RECT_YZ, y0, y1, z0, z1);
impl<M: Material + LowerMaterial> Lower for XYRect<M> { fn lower(&self, l: &mut Lowerer) { self.push_rect(l) } }
impl<M: Material + LowerMaterial> Lower for YZRect<M> { fn lower(&self, l: &mut Lowerer) { self.push_rect(l) } }
impl<M: Material + LowerMaterial> Lower for XZRect<M> {
    fn lower(&self, l: &mut Lowerer) { self.push_rect(l) }
    fn lower_light(&self, l: &mut Lowerer) { // the only rect with pdf_value / random of its own (src/objects.rs:292-309)
        l.lights.push(mirt_light { kind: MIRT_LIGHT_XZ_RECT, p: [self.x0, self.x1, self.z0, self.z1, self.k] });
    }
}
impl<M: Material + LowerMaterial> Lower for Box3d<M> {
    // one record: the device tests the six rects in the reference's order (src/objects.rs:396-436); all six share
    // the material Box3d::new cloned into them
    fn lower(&self, l: &mut Lowerer) {
        let m = self.xy_min.material.lower(l);
        let h = l.header(m);
        l.boxes.push(mirt_box { h, min: v3(&self.min), max: v3(&self.max) });
        l.note_boundary(MIRT_PRIM_BOX, l.boxes.len() as u32 - 1);
    }
}
impl<T: Hittable + Lower> Lower for Translate<T> {
    fn lower(&self, l: &mut Lowerer) {
        l.wrappers.push(mirt_chain_op { kind: MIRT_OP_TRANSLATE, p: v3(&self.offset) });
        self.object.lower(l);
        l.wrappers.pop();
    }
}
impl<T: Hittable + Lower> Lower for Rotate<T> {
    fn lower(&self, l: &mut Lowerer) { // the sine and cosine Rotate::new cached (src/transforms.rs:41-43)
        l.wrappers.push(mirt_chain_op { kind: MIRT_OP_ROTATE_Y, p: [self.sin_theta, self.cos_theta, 0.0] });
        self.object.lower(l);
        l.wrappers.pop();
    }
}
impl<T: Hittable + Lower> Lower for FlipFace<T> {
    fn lower(&self, l: &mut Lowerer) {
        l.wrappers.push(mirt_chain_op { kind: MIRT_OP_FLIP_FACE, p: [0.0; 3] });
        self.object.lower(l);
        l.wrappers.pop();
    }
}
impl<T: Hittable + Lower, M: Material + LowerMaterial> Lower for ConstantMedium<T, M> {
    fn lower(&self, l: &mut Lowerer) {
        if l.boundary_mode { l.unsupported("a constant medium used as the boundary of another medium"); return; }
        let material = self.phase_function.lower(l);
        let mut m = mirt_medium { neg_inv_density: self.neg_inv_density, material, chain: l.current_chain(), order: l.visit_counter,
                                  type_index: l.type_index, object_index: l.object_index, ..Default::default() };
        l.visit_counter += 1;
        // the boundary is lowered as a primitive the tree does not hold (MIRT_PRIM_FLAG_BOUNDARY)
        l.boundary_mode = true; l.boundary_prims = 0; l.last_boundary = None;
        let keep = l.visit_counter;
        self.boundary.lower(l);
        l.visit_counter = keep;
        l.boundary_mode = false;
        match (l.boundary_prims, l.last_boundary) {
            (1, Some((tag, index))) => { m.boundary_tag = tag; m.boundary_index = index; l.media.push(m); }
            _ => l.unsupported("constant medium whose boundary is not a single primitive"),
        }
    }
}

/// A World forgets the types of its members (BlobVec, src/blobvec.rs): `World::add` has to keep, next to each
/// column, a way to lower its elements -- shown here as `lowerers: Vec<fn(&BlobVec, usize, &mut Lowerer, bool)>`,
/// filled in `add::<T>` with a function that casts `blob.get(i)` back to `&T` exactly as HittableVTable::new::<T>
/// does for `hit` (src/hittable.rs:68-79).  The last argument selects lower() or lower_light().
impl Lower for World {
    /// Nested inside another object: members keep the enclosing top-level WorldIndex.
    fn lower(&self, l: &mut Lowerer) {
        for (column, (_vtable, blob)) in self.data.iter().enumerate() {
            for i in 0..blob.len() { (self.lowerers[column])(blob, i, l, false); }
        }
    }
    /// As the `lights` argument: one group per concrete type, because World::random picks a type first and then an
    /// object of it (src/world.rs:112-117) while World::pdf_value averages over all objects (:100-110).
    fn lower_lights_argument(&self, l: &mut Lowerer) {
        l.lights_mode = MIRT_LIGHTS_WORLD;
        for (column, (_vtable, blob)) in self.data.iter().enumerate() {
            let group = mirt_light_group { first_light: l.lights.len() as u32, n_lights: blob.len() as u32 };
            for i in 0..blob.len() { (self.lowerers[column])(blob, i, l, true); }
            l.light_groups.push(group);
        }
    }
}
impl World {
    /// As the scene root: every member is a top-level object with a WorldIndex of its own (src/world.rs:13-17).
    pub(crate) fn lower_root(&self, l: &mut Lowerer) {
        for (column, (_vtable, blob)) in self.data.iter().enumerate() {
            for i in 0..blob.len() {
                l.type_index = column as u32;
                l.object_index = i as u32;
                (self.lowerers[column])(blob, i, l, false);
            }
        }
    }
}
impl BVH {
    /// Members are visited in the left-to-right order of the leaves of the tree create_node built
    /// (src/bvh.rs:31-81): that order is what decides between hits at the same t (`ties go to the right-hand leaf`,
    /// :118-131).  The device builds its own acceleration structure; the reference's nodes are only read for this.
    pub(crate) fn lower_root(&self, l: &mut Lowerer) {
        fn leaves(nodes: &[BVHNode2], index: &BVHNIndex, out: &mut Vec<WorldIndex>) {
            match index {
                BVHNIndex::Node(n) => { leaves(nodes, &nodes[*n].left, out); leaves(nodes, &nodes[*n].right, out); }
                BVHNIndex::WorldIndex((wi, _aabb)) => out.push(*wi),
            }
        }
        let mut order = Vec::new();
        if let Some(root) = self.nodes.len().checked_sub(1) { leaves(&self.nodes, &BVHNIndex::Node(root), &mut order); }
        for wi in order {
            l.type_index = wi.type_index as u32;
            l.object_index = wi.object_index as u32;
            let (_vtable, blob) = &self.world.data[wi.type_index];
            (self.world.lowerers[wi.type_index])(blob, wi.object_index, l, false);
        }
        // media keep the numbering of a column-order walk whatever order the leaves were visited in
        l.media.sort_by_key(|m| (m.type_index, m.object_index));
    }
}
