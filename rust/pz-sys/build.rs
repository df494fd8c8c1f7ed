// PZ_LIB_DIR = the directory holding libpz_hip.so (paillier_halo2_amd/csrc of this repository after `make`)
fn main() {
    if let Ok(dir) = std::env::var("PZ_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=pz_hip");
    println!("cargo:rerun-if-env-changed=PZ_LIB_DIR");
}
