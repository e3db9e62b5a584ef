// Links libbpg.so, built beforehand by `bash build.sh` at the repository root (hipcc, gfx950).
// BPG_LIB_DIR: the directory that holds libbpg.so (default: ../../bulletproofs_gadgets_amd).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("BPG_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../bulletproofs_gadgets_amd")
    });
    let dir = dir.canonicalize().unwrap_or(dir);
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=bpg");
    // run-time search path, so that `cargo test` finds the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=BPG_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/bpg.h");
}
