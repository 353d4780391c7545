// build.rs for zkir-runtime with the MI355X back-end (drop next to zkir-runtime/Cargo.toml and add `build = "build.rs"` plus the `gpu` feature below).
//
// NOT COMPILED IN THIS REPOSITORY: the build image has no Rust toolchain (no rustc / cargo, crates.io unreachable) — this file and gpu.rs are the binding a
// maintainer of seceq/zkir would add, written against include/zkir_amd.h; tests/cpp/reference_tests.cpp exercises the same C ABI calls from C++.
//
//   [features]
//   gpu = []
//
// ZKIR_AMD_LIB_DIR = the directory that holds libzkir_amd.so (built by `python -m zkir_amd.build`, i.e. hipcc --offload-arch=gfx950).
fn main() {
    if std::env::var_os("CARGO_FEATURE_GPU").is_some() {
        let dir = std::env::var("ZKIR_AMD_LIB_DIR").expect("set ZKIR_AMD_LIB_DIR to the directory of libzkir_amd.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=zkir_amd");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
        println!("cargo:rerun-if-env-changed=ZKIR_AMD_LIB_DIR");
    }
}
