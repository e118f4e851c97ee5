// Tells cargo where librten_hip.so lives.  The library is built by `rten_amd/csrc/build.sh` (hipcc, gfx950 only);
// RTEN_HIP_LIB_DIR overrides the in-tree location.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("RTEN_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../rten_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rten_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=RTEN_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/rten_hip.h");
}
