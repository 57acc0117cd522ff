// rust/src/gates_hip.rs -- the batch entry points of src/gates.rs:352-547 and src/trgsw.rs:289-294, and Gates::mux /
// mux_naive (gates.rs:157-199), on the GPU.  Copy to `src/gates_hip.rs` of the crate, add
// `#[cfg(feature = "hip")] pub mod gates_hip;` to src/lib.rs, and apply rust/patches/*.patch, which make
// `gates::batch_*`, `trgsw::batch_blind_rotate` and `default_bootstrap()` call these under `--features hip`.
// UNCOMPILED (no Rust toolchain in this image).  Every function keeps the SIGNATURE of its namesake in the reference
// (`&[(Ciphertext, Ciphertext)], &CloudKey -> Vec<Ciphertext>`; tests/test_binding_lint.py holds each one to
// /root/reference/src/gates.rs when the reference tree is present): the engine is the process-wide one
// (`bootstrap::hip::default_engine()`), as the Railgun of the reference's functions is `default_railgun()`.
//
// One call = prep + blind rotate + sample extract + key switch for the whole slice, cut contiguously over the GPUs of
// the engine, results in input order (rayon_impl.rs:40-47 `par_iter().map().collect()`), bit-identical to the CPU
// functions (rust/tests/hip_gates.rs asserts that word for word).
use crate::bootstrap::hip::{self, default_engine};
use crate::key::CloudKey;
use crate::tlwe;
use crate::trlwe;
use crate::utils::Ciphertext;

/// gates.rs:352-383
pub fn batch_nand_hip(inputs: &[(Ciphertext, Ciphertext)], cloud_key: &CloudKey) -> Vec<Ciphertext> {
    default_engine().batch_gate(hip::NAND, inputs, cloud_key)
}

/// gates.rs:388-418
pub fn batch_and_hip(inputs: &[(Ciphertext, Ciphertext)], cloud_key: &CloudKey) -> Vec<Ciphertext> {
    default_engine().batch_gate(hip::AND, inputs, cloud_key)
}

/// gates.rs:423-450
pub fn batch_or_hip(inputs: &[(Ciphertext, Ciphertext)], cloud_key: &CloudKey) -> Vec<Ciphertext> {
    default_engine().batch_gate(hip::OR, inputs, cloud_key)
}

/// gates.rs:455-482
pub fn batch_xor_hip(inputs: &[(Ciphertext, Ciphertext)], cloud_key: &CloudKey) -> Vec<Ciphertext> {
    default_engine().batch_gate(hip::XOR, inputs, cloud_key)
}

/// gates.rs:487-514
pub fn batch_nor_hip(inputs: &[(Ciphertext, Ciphertext)], cloud_key: &CloudKey) -> Vec<Ciphertext> {
    default_engine().batch_gate(hip::NOR, inputs, cloud_key)
}

/// gates.rs:519-547
pub fn batch_xnor_hip(inputs: &[(Ciphertext, Ciphertext)], cloud_key: &CloudKey) -> Vec<Ciphertext> {
    default_engine().batch_gate(hip::XNOR, inputs, cloud_key)
}

/// trgsw.rs:289-294: one blind rotation per ciphertext with the cloud key's own test vector
pub fn batch_blind_rotate_hip(srcs: &[tlwe::TLWELv0], cloud_key: &CloudKey) -> Vec<trlwe::TRLWELv1> {
    default_engine().batch_blind_rotate(srcs, cloud_key)
}

/// Gates::mux in the reference's formula (gates.rs:157-183: two bootstrap_without_key_switch, their sum, one bootstrap)
pub fn mux_hip(tlwe_a: &Ciphertext, tlwe_b: &Ciphertext, tlwe_c: &Ciphertext, cloud_key: &CloudKey) -> Ciphertext {
    default_engine().batch_mux(false, &[(*tlwe_a, *tlwe_b, *tlwe_c)], cloud_key).pop().unwrap()
}

/// Gates::mux_naive (gates.rs:189-199): and(a, b), and(not(a), c), or
pub fn mux_naive_hip(tlwe_a: &Ciphertext, tlwe_b: &Ciphertext, tlwe_c: &Ciphertext, cloud_key: &CloudKey) -> Ciphertext {
    default_engine().batch_mux(true, &[(*tlwe_a, *tlwe_b, *tlwe_c)], cloud_key).pop().unwrap()
}

/// The batched form the GPU is meant to be fed with (no counterpart in the reference): many muxes in one call.
pub fn batch_mux_hip(inputs: &[(Ciphertext, Ciphertext, Ciphertext)], cloud_key: &CloudKey) -> Vec<Ciphertext> {
    default_engine().batch_mux(false, inputs, cloud_key)
}

/// ... and mux_naive
pub fn batch_mux_naive_hip(inputs: &[(Ciphertext, Ciphertext, Ciphertext)], cloud_key: &CloudKey) -> Vec<Ciphertext> {
    default_engine().batch_mux(true, inputs, cloud_key)
}
