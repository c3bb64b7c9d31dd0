// Link libotters_hip.so.  The directory that holds it comes from OTTERS_HIP_LIB_DIR (e.g. <repo>/otters_amd/csrc); the
// library itself dlopen()s librccl.so.1 on first use of the multi-GPU entry points, so nothing else is linked here.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=OTTERS_HIP_LIB_DIR");
    let dir = match env::var("OTTERS_HIP_LIB_DIR") {
        Ok(d) => PathBuf::from(d),
        Err(_) => {
            // default: the in-tree build of this repository, three levels up from the crate
            let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
            manifest.join("..").join("..").join("..").join("otters_amd").join("csrc")
        }
    };
    if !dir.join("libotters_hip.so").exists() {
        panic!(
            "libotters_hip.so not found in {}: build it (`python -c 'import __graft_entry__ as g; g.build()'` or \
             `make -C otters_amd/csrc`) or point OTTERS_HIP_LIB_DIR at the directory that holds it",
            dir.display()
        );
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=otters_hip");
    // so that `cargo test` / `cargo run` find the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
