// Source of the `hip-vocoder` Cargo feature's Rust side.  Written against the reference tree;
// NOT compiled here (this image has no cargo / rustc).

// goes into fn main() of build.rs (pattern of build.rs:26-38,158-195)
#[cfg(feature = "hip-vocoder")]
{
    // hipvoc/ = this repository's qwen3-tts_amd/csrc + include, vendored or a submodule
    let status = std::process::Command::new("make")
        .args(["-C", "hipvoc/qwen3-tts_amd/csrc", "../libhipvoc.a", "ARCH=gfx950"])
        .status().expect("hipcc build of libhipvoc failed to start");
    assert!(status.success(), "libhipvoc build failed");
    println!("cargo:rustc-link-search=native=hipvoc/qwen3-tts_amd");
    println!("cargo:rustc-link-lib=static=hipvoc");
    println!("cargo:rustc-link-search=native=/opt/rocm/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-lib=dylib=stdc++");
    println!("cargo:rerun-if-changed=hipvoc/qwen3-tts_amd/csrc");
}
