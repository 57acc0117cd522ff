#!/bin/sh
# Put the MI355X binding into a checkout of thedonutfactory/rs-tfhe (v0.2.0):   rust/apply.sh /path/to/rs-tfhe
# Copies the three source files and the test, applies rust/patches/rs-tfhe-hip.patch (the `hip` feature, the build.rs
# link block, `pub mod hip` / `pub mod gates_hip`, the `rows()` accessor, and the cfg(feature = "hip") routes of
# default_bootstrap(), gates::batch_* and trgsw::batch_blind_rotate).  Then, on a box with an MI355X:
#   make -C <this repo>/rs-tfhe_amd/csrc
#   TFHE_HIP_LIB_DIR=<this repo>/rs-tfhe_amd LD_LIBRARY_PATH=$TFHE_HIP_LIB_DIR cargo test --release --features "hip lut-bootstrap proxy-reenc"
set -e
crate=${1:?path to the rs-tfhe checkout}
here=$(cd "$(dirname "$0")" && pwd)
cp "$here/src/bootstrap/hip.rs" "$crate/src/bootstrap/hip.rs"
cp "$here/src/gates_hip.rs" "$crate/src/gates_hip.rs"
cp "$here/src/proxy_reenc_hip.rs" "$crate/src/proxy_reenc_hip.rs"
mkdir -p "$crate/tests"
cp "$here/tests/hip_gates.rs" "$crate/tests/hip_gates.rs"
patch -d "$crate" -p1 < "$here/patches/rs-tfhe-hip.patch"
echo "applied: build with --features hip (TFHE_HIP_LIB_DIR = directory of libtfhe_hip.so)"
