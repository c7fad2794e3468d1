// Builds the HIP kernels for gfx950 (hipcc --offload-arch=gfx950, see the Makefile) and links the C ABI.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let csrc = PathBuf::from(env::var("RT_MI355X_SRC").expect("set RT_MI355X_SRC to raytracer_amd/csrc"));
    let status = Command::new("make")
        .arg("-C")
        .arg(&csrc)
        .arg("librt_mi355x.so")
        .status()
        .expect("failed to run make");
    assert!(status.success(), "building librt_mi355x.so failed");
    println!("cargo:rustc-link-search=native={}", csrc.display());
    println!("cargo:rustc-link-lib=dylib=rt_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", csrc.display());
    for f in ["kernels.hip", "api.cpp", "scene_build.cpp", "gltf.cpp", "png_decode.cpp", "Makefile"] {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
    println!("cargo:rerun-if-env-changed=RT_MI355X_SRC");
}
