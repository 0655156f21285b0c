// Tells rustc where libzbra_hip.so lives.  ZBRA_HIP_LIB_DIR names the directory that holds it
// (the repository's zbra_amd/ after `make -C zbra_amd/csrc`); without it the default library search
// path is used.  The library itself links the HIP runtime; nothing else is needed here.
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=ZBRA_HIP_LIB_DIR");
    if let Ok(dir) = env::var("ZBRA_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        // so that test binaries and the CLI find it without LD_LIBRARY_PATH
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=zbra_hip");
}
