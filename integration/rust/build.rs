// build.rs of the crate when the `mi355x` feature is on: tell rustc where libblsbn254_hip.so lives.
// BLSBN254_LIB_DIR = the engine checkout's `bls-bn254_amd/` directory (built with `make -C bls-bn254_amd`).
fn main() {
    if std::env::var_os("CARGO_FEATURE_MI355X").is_some() {
        let dir = std::env::var("BLSBN254_LIB_DIR").expect("set BLSBN254_LIB_DIR to the directory holding libblsbn254_hip.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
        println!("cargo:rerun-if-env-changed=BLSBN254_LIB_DIR");
    }
}
