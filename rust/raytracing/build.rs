// Points the linker at librt_hip.so.  RT_HIP_LIB_DIR overrides the in-tree location (<repo>/raytracing_amd/csrc, where
// `make -C raytracing_amd/csrc` leaves the library).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("RT_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
        manifest.join("..").join("..").join("raytracing_amd").join("csrc")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=RT_HIP_LIB_DIR");
}
