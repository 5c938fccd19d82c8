// Links libptcore.so.  PTCORE_LIB_DIR names the directory that holds it (pathtracer_amd/ of the
// repository after `make`); the same directory goes into the binary's rpath.
fn main() {
    println!("cargo:rerun-if-env-changed=PTCORE_LIB_DIR");
    if let Ok(dir) = std::env::var("PTCORE_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=ptcore");
}
