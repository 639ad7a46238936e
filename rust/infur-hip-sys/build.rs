// UNTESTED (no Rust toolchain in the build environment).
fn main() {
    // INFUR_HIP_LIB_DIR = directory holding libinfur_hip.so (infur_amd/ in the source tree)
    if let Ok(dir) = std::env::var("INFUR_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=infur_hip");
}
