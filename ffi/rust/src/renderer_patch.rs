//! The new body of `Renderer::render` (src/renderer.rs:43-92): lower, upload, render, copy into `self.buffer`.
//! Source only, never compiled (README.md).  Tested twin: `rt::Renderer::render` in raytracing_amd/host/rt.hpp.
//! `present()` (src/renderer.rs:94-138) is unchanged: `self.buffer` has exactly the layout mirt_render writes.
use crate::ffi::*;
use crate::gpu::Context;
use crate::lower::{Lower, Lowerer};

impl Renderer {
    pub fn render<H, L>(&mut self, hittable: &H, camera: &Camera, lights: Option<&L>) -> Result<(), String>
    where
        H: Hittable + LowerRoot + Sync,
        L: Hittable + Lower + Sync,
    {
        let now = std::time::Instant::now();
        let mut l = Lowerer::new();
        hittable.lower_root(&mut l); // World: one WorldIndex per member; BVH: leaf order; anything else: (0, 0)
        if let Some(lights) = lights { lights.lower_lights_argument(&mut l); }
        if let Some(why) = l.error.take() { return Err(format!("cannot lower scene: {}", why)); }
        let desc = l.desc(camera.time0, camera.time1);
        // every GPU of the node, as the reference takes every hardware thread (src/renderer.rs:56); MIRT_DEVICES="0"
        // or "0,1,2,3" narrows it
        let devices: Vec<i32> = std::env::var("MIRT_DEVICES").ok()
            .map(|s| s.split(',').filter_map(|d| d.trim().parse().ok()).collect())
            .unwrap_or_else(|| vec![0]);
        let ctx = Context::over(&devices, 0)?;
        let scene = ctx.upload(&desc)?;
        let cam = mirt_camera {
            origin: [camera.origin.x, camera.origin.y, camera.origin.z],
            lower_left_corner: [camera.lower_left_corner.x, camera.lower_left_corner.y, camera.lower_left_corner.z],
            horizontal: [camera.horizontal.x, camera.horizontal.y, camera.horizontal.z],
            vertical: [camera.vertical.x, camera.vertical.y, camera.vertical.z],
            u: [camera.u.x, camera.u.y, camera.u.z], v: [camera.v.x, camera.v.y, camera.v.z], w: [camera.w.x, camera.w.y, camera.w.z],
            lens_radius: camera.lens_radius, time0: camera.time0, time1: camera.time1,
        };
        // MIRT_STRIP_THREADS=t: the frame as the present code leaves it on t hardware threads (src/renderer.rs:56-71):
        // height % t rows never rendered (black, at the end of the buffer), the image that many rows higher.
        // Unset: every row is rendered.  C++ twin: rt::Renderer::set_strip_threads / rt::ReferenceStrips.
        let strips: u32 = std::env::var("MIRT_STRIP_THREADS").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        if strips > self.screen_height { return Err("more strip threads than rows".into()); }
        let kept = if strips == 0 { self.screen_height } else { strips * (self.screen_height / strips) };
        let params = mirt_render_params {
            width: self.screen_width, height: self.screen_height, samples_per_pixel: self.samples_per_pixel,
            first_sample: 0, max_depth: self.max_depth,
            background: [self.background.x, self.background.y, self.background.z],
            seed: rand::random(), path_pool: 0, flags: 0,
            first_row: self.screen_height - kept, n_rows: if strips == 0 { 0 } else { kept },
        };
        scene.render(&cam, &params, &mut self.buffer)?;
        if kept < self.screen_height {
            let pitch = self.screen_width as usize * 3;
            self.buffer.copy_within(pitch * (self.screen_height - kept) as usize.., 0);
            self.buffer[pitch * kept as usize..].fill(0);
        }
        println!("Rendered in {}ms", now.elapsed().as_millis());
        Ok(())
    }
}

/// What the scene root is: a World, a BVH, or any single object.
pub(crate) trait LowerRoot { fn lower_root(&self, l: &mut Lowerer); }
impl LowerRoot for World { fn lower_root(&self, l: &mut Lowerer) { World::lower_root(self, l) } }
impl LowerRoot for BVH { fn lower_root(&self, l: &mut Lowerer) { BVH::lower_root(self, l) } }
