// Points the linker at libplonk_hip.so.  PLONK_HIP_LIB_DIR names the directory that holds it
// (default: plonk_amd/ of this repository, where `make -C plonk_amd/csrc` writes it).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("PLONK_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../plonk_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=plonk_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=PLONK_HIP_LIB_DIR");
}
