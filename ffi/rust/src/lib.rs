//! Rust side of the MI355X path-tracing backend.  Source only, never compiled: see README.md.
pub mod ffi;
pub mod gpu;
// `lower.rs` and `renderer_patch.rs` belong inside the reference crate (they read its private fields).
