//! Links libdehalo.so.  DEHALO_LIB_DIR = the directory that holds it (the repository's `delay-encryption-in-halo2_amd/` after `make`); the library
//! itself needs the ROCm runtime (libamdhip64.so) on the loader path at run time.  Nothing is compiled here: the kernels are built by the repository's
//! Makefile with hipcc for gfx950, and `src/lib.rs` is generated from `include/dehalo.h` by `tools/gen_rust_bindings.py`.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=DEHALO_LIB_DIR");
    let dir = env::var("DEHALO_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        // default: this crate lives at <repo>/bindings/rust/dehalo-sys
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../delay-encryption-in-halo2_amd")
    });
    assert!(dir.join("libdehalo.so").exists(), "libdehalo.so not found in {} (run `make` in the repository root, or set DEHALO_LIB_DIR)", dir.display());
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=dehalo");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,/opt/rocm/lib");
}
