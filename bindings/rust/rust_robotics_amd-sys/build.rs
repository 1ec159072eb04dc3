// Points the linker at librust_robotics_amd.so.  RUST_ROBOTICS_AMD_LIB_DIR overrides the default
// location (the in-tree build output of `make -C rust_robotics_amd/csrc`).
fn main() {
    let dir = std::env::var("RUST_ROBOTICS_AMD_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../../../rust_robotics_amd", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rust_robotics_amd");
    println!("cargo:rerun-if-env-changed=RUST_ROBOTICS_AMD_LIB_DIR");
}
