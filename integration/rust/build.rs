// Links libheat_amd.so (built by `python -m heat_amd.build`); HEAT_AMD_LIB_DIR = <repo>/heat_amd/lib.
fn main() {
    if let Ok(dir) = std::env::var("HEAT_AMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=heat_amd");
}
