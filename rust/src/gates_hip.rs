// rust/src/gates_hip.rs -- batch gates, blind rotation and mux on the GPU: paste next to `batch_nand_with_railgun` in
// src/gates.rs (:357-383) and `batch_blind_rotate` in src/trgsw.rs (:289).  UNCOMPILED (no Rust toolchain in this image).
use crate::bootstrap::hip::HipEngine;
use crate::key::CloudKey;
use crate::trlwe;
use crate::utils::Ciphertext;

#[cfg(feature = "hip")]
pub fn batch_nand_hip(inputs: &[(Ciphertext, Ciphertext)], cloud_key: &CloudKey, engine: &HipEngine) -> Vec<Ciphertext> {
    engine.batch_gate(crate::bootstrap::hip::NAND, inputs, cloud_key)   // prep + blind rotate + extract + key switch, every device
}
// batch_and_hip / batch_or_hip / batch_xor_hip / batch_nor_hip / batch_xnor_hip: same with AND/OR/XOR/NOR/XNOR
#[cfg(feature = "hip")]
pub fn batch_blind_rotate_hip(srcs: &[Ciphertext], cloud_key: &CloudKey, engine: &HipEngine) -> Vec<trlwe::TRLWELv1> {
    engine.batch_blind_rotate(srcs, cloud_key)
}
#[cfg(feature = "hip")]
pub fn mux_hip(a: &Ciphertext, b: &Ciphertext, c: &Ciphertext, cloud_key: &CloudKey, engine: &HipEngine) -> Ciphertext {
    engine.batch_mux(false, &[(a.clone(), b.clone(), c.clone())], cloud_key).pop().unwrap()   // Gates::mux, gates.rs:157-183
}
