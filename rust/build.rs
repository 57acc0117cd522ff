// rust/build.rs -- the block to add to the crate's build.rs, next to the existing SPQLIOS block (build.rs:7-23 of the
// reference).  UNCOMPILED (no Rust toolchain in this image).
fn main() {
    if std::env::var("CARGO_FEATURE_HIP").is_ok() {
        let dir = std::env::var("TFHE_HIP_LIB_DIR").expect("set TFHE_HIP_LIB_DIR to the directory of libtfhe_hip.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=tfhe_hip");
    }
}
