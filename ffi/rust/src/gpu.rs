//! Safe handles over the C ABI.  Every status other than MIRT_OK becomes `Err(String)` with the library's message,
//! which is what `Renderer::render` returns (`Result<(), String>`, src/renderer.rs:43-52).
use crate::ffi::*;
use std::ffi::CStr;
use std::ptr;

pub struct Context { raw: *mut mirt_context }
pub struct Scene<'c> { ctx: &'c Context, raw: *mut mirt_scene }
pub struct Frame<'c> { ctx: &'c Context, raw: *mut mirt_frame }

// A context serves one call at a time; it may move between threads but is not shared.
unsafe impl Send for Context {}

fn create_error() -> String {
    unsafe { CStr::from_ptr(mirt_last_error(ptr::null())).to_string_lossy().into_owned() }
}

/// The structs of ffi.rs mirror include/mirt.h at this version and carry no size field: an older library
/// would read and write past them.
const HEADER_VERSION: i32 = 200;
fn check_version() -> Result<(), String> {
    let v = unsafe { mirt_version() };
    if v >= HEADER_VERSION { Ok(()) } else { Err(format!("libmirt {} is older than the bindings ({})", v, HEADER_VERSION)) }
}

impl Context {
    /// One GPU.
    pub fn new(device: i32) -> Result<Self, String> {
        check_version()?;
        let mut raw = ptr::null_mut();
        if unsafe { mirt_context_create(device, &mut raw) } != MIRT_OK { return Err(create_error()); }
        Ok(Self { raw })
    }
    /// Several GPUs of the node: the frame is shared out between them as the reference shares it out between
    /// threads (src/renderer.rs:55-87).  `row_bands` = 0: sample ranges only.
    pub fn over(devices: &[i32], row_bands: u32) -> Result<Self, String> {
        check_version()?;
        let mut raw = ptr::null_mut();
        if unsafe { mirt_context_create_multi(devices.as_ptr(), devices.len() as i32, row_bands, &mut raw) } != MIRT_OK {
            return Err(create_error());
        }
        Ok(Self { raw })
    }
    pub fn raw(&self) -> *mut mirt_context { self.raw }
    pub fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(mirt_last_error(self.raw)).to_string_lossy().into_owned() }
    }
    pub(crate) fn check(&self, status: i32) -> Result<(), String> {
        if status == MIRT_OK { Ok(()) } else { Err(self.last_error()) }
    }
    pub fn upload<'c>(&'c self, desc: &mirt_scene_desc) -> Result<Scene<'c>, String> {
        let mut raw = ptr::null_mut();
        self.check(unsafe { mirt_scene_upload(self.raw, desc, &mut raw) })?;
        Ok(Scene { ctx: self, raw })
    }
}
impl Drop for Context { fn drop(&mut self) { unsafe { mirt_context_destroy(self.raw) } } }

impl<'c> Scene<'c> {
    pub fn raw(&self) -> *const mirt_scene { self.raw }
    /// The body of Renderer::render: `rgb24` is Renderer::buffer (W*H*3 bytes, row 0 = image top, src/renderer.rs:254-258).
    pub fn render(&self, camera: &mirt_camera, params: &mirt_render_params, rgb24: &mut [u8]) -> Result<mirt_stats, String> {
        let need = params.width as usize * params.height as usize * 3;
        if rgb24.len() != need { return Err(format!("buffer holds {} bytes, the frame needs {}", rgb24.len(), need)); }
        let mut stats = mirt_stats::default();
        self.ctx.check(unsafe { mirt_render(self.ctx.raw, self.raw, camera, params, rgb24.as_mut_ptr(), ptr::null_mut(), &mut stats) })?;
        Ok(stats)
    }
}
impl<'c> Drop for Scene<'c> { fn drop(&mut self) { unsafe { mirt_scene_destroy(self.ctx.raw, self.raw) } } }

impl<'c> Frame<'c> {
    pub fn new(ctx: &'c Context, width: u32, height: u32) -> Result<Self, String> {
        let mut raw = ptr::null_mut();
        ctx.check(unsafe { mirt_frame_create(ctx.raw, width, height, &mut raw) })?;
        Ok(Self { ctx, raw })
    }
    pub fn samples(&self) -> u32 { unsafe { mirt_frame_samples(self.raw) } }
    /// Adds `params.samples_per_pixel` more samples per pixel.
    pub fn render(&mut self, scene: &Scene, camera: &mirt_camera, params: &mirt_render_params) -> Result<mirt_stats, String> {
        let mut stats = mirt_stats::default();
        self.ctx.check(unsafe { mirt_frame_render(self.ctx.raw, scene.raw, camera, params, self.raw, &mut stats) })?;
        Ok(stats)
    }
    /// The frame as it stands: what `present()` (src/renderer.rs:94-138) would show now.
    pub fn read(&self, rgb24: &mut [u8]) -> Result<(), String> {
        self.ctx.check(unsafe { mirt_frame_read(self.ctx.raw, self.raw, rgb24.as_mut_ptr(), ptr::null_mut()) })
    }
    pub fn save(&self, path: &str) -> Result<(), String> {
        let c = std::ffi::CString::new(path).map_err(|e| e.to_string())?;
        self.ctx.check(unsafe { mirt_frame_save(self.ctx.raw, self.raw, c.as_ptr()) })
    }
}
impl<'c> Drop for Frame<'c> { fn drop(&mut self) { unsafe { mirt_frame_destroy(self.ctx.raw, self.raw) } } }
