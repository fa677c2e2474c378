// build.rs of the `heat` crate with the `gpu` feature: links libheat_amd.so (built by `python -m heat_amd.build`);
// HEAT_AMD_LIB_DIR = <heat_amd repo>/heat_amd/lib. Without the feature it does nothing.
fn main() {
    if std::env::var("CARGO_FEATURE_GPU").is_err() {
        return;
    }
    if let Ok(dir) = std::env::var("HEAT_AMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=heat_amd");
}
