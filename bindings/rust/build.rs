// build.rs -- link libsoundsym_amd.so and the ONE HIP runtime of the process.
//
// libsoundsym_amd.so deliberately has no DT_NEEDED on libamdhip64 (a process must hold exactly one HIP
// runtime, and hosts such as PyTorch ship their own copy), so a plain Rust binary links ROCm's itself.
// RCCL is NOT linked: the library binds it at run time (dlsym, then librccl.so.1 / $SSYM_RCCL_LIB) and only
// the ssym_comm_* / ssym_match_sharded calls need it.
use std::env;

fn main() {
    let lib_dir = env::var("SOUNDSYM_AMD_LIB_DIR").unwrap_or_else(|_| "../../soundsym_amd".to_string());
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}", lib_dir);
    println!("cargo:rustc-link-lib=dylib=soundsym_amd");
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rerun-if-env-changed=SOUNDSYM_AMD_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
}
