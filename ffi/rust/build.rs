// Links libmirt.so (built by `make -C raytracing_amd/csrc`).  MIRT_LIB_DIR names the directory that holds it.
fn main() {
    let dir = std::env::var("MIRT_LIB_DIR").unwrap_or_else(|_| "../../raytracing_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=mirt");
    println!("cargo:rerun-if-env-changed=MIRT_LIB_DIR");
}
