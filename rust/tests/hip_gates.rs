// rust/tests/hip_gates.rs -- copy to `tests/hip_gates.rs` of the crate; runs with
//     TFHE_HIP_LIB_DIR=/path/to/rs-tfhe_amd LD_LIBRARY_PATH=$TFHE_HIP_LIB_DIR cargo test --features "hip lut-bootstrap" --test hip_gates
// on a box with an MI355X.  UNCOMPILED (no Rust toolchain in this image).
//
// The reference's own unit tests for this path, restated against the GPU strategies: src/gates.rs:553-681 and :785-856
// (every gate's truth table through `Gates`, mux_naive, a custom strategy), src/bootstrap/vanilla.rs:78-131,
// src/bootstrap/lut.rs:142-279, src/trgsw.rs:507-529 (blind rotate + sample extract decrypts) -- plus what the reference
// cannot assert about itself: the GPU result equals the CPU result WORD FOR WORD (the f64 products are exact at these
// parameters, so there is one right answer), for single calls, batches, and batches larger than one device's shard.
#![cfg(feature = "hip")]

use rand::Rng;
use rs_tfhe::bootstrap::hip::{default_engine, HipBootstrap};
use rs_tfhe::bootstrap::vanilla::VanillaBootstrap;
use rs_tfhe::bootstrap::{default_bootstrap, Bootstrap};
use rs_tfhe::gates::{self, Gates};
use rs_tfhe::gates_hip;
use rs_tfhe::key::{CloudKey, SecretKey};
use rs_tfhe::parallel::default_railgun;
use rs_tfhe::utils::Ciphertext;
use rs_tfhe::{params, tlwe, trgsw, trlwe};

fn keys() -> (SecretKey, CloudKey) {
    let key = SecretKey::new();
    let cloud_key = CloudKey::new(&key);
    (key, cloud_key)
}

fn enc(b: bool, key: &SecretKey) -> Ciphertext {
    Ciphertext::encrypt_bool(b, params::tlwe_lv0::ALPHA, &key.key_lv0)
}

/// gates.rs:832-856, with the GPU strategy injected the way gates.rs:43-45 provides
fn test_gate<E: Fn(bool, bool) -> bool, C: Fn(&Gates, &Ciphertext, &Ciphertext, &CloudKey) -> Ciphertext>(expect: E, actual: C) {
    let (key, cloud_key) = keys();
    let gates = Gates::with_bootstrap(Box::new(HipBootstrap::new()));
    for (a, b) in [(true, true), (true, false), (false, true), (false, false)] {
        let (ct_a, ct_b) = (enc(a, &key), enc(b, &key));
        let result = actual(&gates, &ct_a, &ct_b, &cloud_key);
        assert_eq!(result.decrypt_bool(&key.key_lv0), expect(a, b), "Failed for {} {}", a, b);
    }
}

#[test] fn test_hom_nand() { test_gate(|a, b| !(a & b), |g, a, b, k| g.nand(a, b, k)); }            // gates.rs:559
#[test] fn test_hom_or() { test_gate(|a, b| a | b, |g, a, b, k| g.or(a, b, k)); }                   // :567
#[test] fn test_hom_xnor() { test_gate(|a, b| false ^ (b ^ a), |g, a, b, k| g.xnor(a, b, k)); }     // :575 (the reference's xnor IS xor)
#[test] fn test_hom_xor() { test_gate(|a, b| a ^ b, |g, a, b, k| g.xor(a, b, k)); }                 // :583
#[test] fn test_hom_not() { test_gate(|a, _| !a, |g, a, _, _| g.not(a)); }                          // :591
#[test] fn test_hom_copy() { test_gate(|a, _| a, |g, a, _, _| g.copy(a)); }                         // :599
#[test] fn test_hom_constant() { test_gate(|_, _| true, |g, _, _, _| g.constant(true)); }           // :607
#[test] fn test_hom_nor() { test_gate(|a, b| !(a | b), |g, a, b, k| g.nor(a, b, k)); }              // :616
#[test] fn test_hom_and_ny() { test_gate(|a, b| !a & b, |g, a, b, k| g.and_ny(a, b, k)); }          // :624
#[test] fn test_hom_and_yn() { test_gate(|a, b| a & !b, |g, a, b, k| g.and_yn(a, b, k)); }          // :632
#[test] fn test_hom_or_ny() { test_gate(|a, b| !a | b, |g, a, b, k| g.or_ny(a, b, k)); }            // :640
#[test] fn test_hom_or_yn() { test_gate(|a, b| a | !b, |g, a, b, k| g.or_yn(a, b, k)); }            // :648

/// gates.rs:656-681: the reference asserts mux_naive only (Gates::mux's formula is not a decryptable construction)
#[test]
fn test_mux() {
    let mut rng = rand::thread_rng();
    let (key, cloud_key) = keys();
    let gates = Gates::with_bootstrap(Box::new(HipBootstrap::new()));
    let cpu = Gates::with_bootstrap(Box::new(VanillaBootstrap::new()));
    for _ in 0..10 {
        let (a, b, c) = (rng.gen::<bool>(), rng.gen::<bool>(), rng.gen::<bool>());
        let (ta, tb, tc) = (enc(a, &key), enc(b, &key), enc(c, &key));
        let op = gates.mux_naive(&ta, &tb, &tc, &cloud_key);
        assert_eq!(op.decrypt_bool(&key.key_lv0), (a & b) | ((!a) & c));
        assert_eq!(op.p, cpu.mux_naive(&ta, &tb, &tc, &cloud_key).p);
        // the one-call forms: same words as the gate-by-gate composition, and Gates::mux's formula bit for bit
        assert_eq!(gates_hip::mux_naive_hip(&ta, &tb, &tc, &cloud_key).p, op.p);
        assert_eq!(gates_hip::mux_hip(&ta, &tb, &tc, &cloud_key).p, cpu.mux(&ta, &tb, &tc, &cloud_key).p);
    }
}

/// gates.rs:785-806
#[test]
fn test_gates_with_custom_bootstrap() {
    let gates = Gates::with_bootstrap(Box::new(HipBootstrap::new()));
    assert_eq!(gates.bootstrap_strategy(), "hip-gfx950");
    let mut rng = rand::thread_rng();
    let (key, cloud_key) = keys();
    let (a, b) = (rng.gen::<bool>(), rng.gen::<bool>());
    let and = gates.and(&enc(a, &key), &enc(b, &key), &cloud_key);
    assert_eq!(and.decrypt_bool(&key.key_lv0), a & b);
}

/// bootstrap/mod.rs:41-43 under `--features hip` (rust/patches/bootstrap_mod.rs.patch)
#[test]
fn test_default_bootstrap_is_the_gpu() {
    assert_eq!(default_bootstrap().name(), "hip-gfx950");
    assert_eq!(Gates::new().bootstrap_strategy(), "hip-gfx950");
}

/// vanilla.rs:78-98, and the same ciphertexts through VanillaBootstrap: identical words
#[test]
fn test_hip_bootstrap() {
    let mut rng = rand::thread_rng();
    let (key, cloud_key) = keys();
    let (gpu, cpu) = (HipBootstrap::new(), VanillaBootstrap::new());
    for _ in 0..10 {
        let plain = rng.gen::<bool>();
        let encrypted = enc(plain, &key);
        let bootstrapped = gpu.bootstrap(&encrypted, &cloud_key);
        assert_eq!(bootstrapped.decrypt_bool(&key.key_lv0), plain);
        assert_eq!(bootstrapped.p, cpu.bootstrap(&encrypted, &cloud_key).p);
    }
}

/// vanilla.rs:100-125 (runs without panicking) -- and here it can be checked: same words as the CPU strategy's
#[test]
fn test_hip_bootstrap_without_key_switch() {
    let mut rng = rand::thread_rng();
    let (key, cloud_key) = keys();
    let (gpu, cpu) = (HipBootstrap::new(), VanillaBootstrap::new());
    for _ in 0..3 {
        let encrypted = enc(rng.gen::<bool>(), &key);
        let intermediate = gpu.bootstrap_without_key_switch(&encrypted, &cloud_key);
        assert_eq!(intermediate.p, cpu.bootstrap_without_key_switch(&encrypted, &cloud_key).p);
    }
}

/// `Bootstrap: Send + Sync` (bootstrap/mod.rs:23): a Rayon team calling ONE strategy -- the user-side counterpart of
/// parallel/rayon_impl.rs:40-47.  The library merges the one-ciphertext calls that are in flight together into shared
/// launches (rs-tfhe_amd/csrc/combine.hpp); every result must still be the CPU strategy's, word for word, in input order,
/// and the team must beat the same calls made one after the other by a wide margin.
#[test]
fn test_a_rayon_team_on_one_strategy() {
    use rayon::prelude::*;
    let mut rng = rand::thread_rng();
    let (key, cloud_key) = keys();
    let gates = Gates::with_bootstrap(Box::new(HipBootstrap::new()));
    let cpu = Gates::with_bootstrap(Box::new(VanillaBootstrap::new()));
    let plains: Vec<(bool, bool)> = (0..512).map(|_| (rng.gen::<bool>(), rng.gen::<bool>())).collect();
    let pairs: Vec<(Ciphertext, Ciphertext)> = plains.iter().map(|&(a, b)| (enc(a, &key), enc(b, &key))).collect();
    let t0 = std::time::Instant::now();
    let one_by_one: Vec<Ciphertext> = pairs.iter().take(32).map(|(a, b)| gates.nand(a, b, &cloud_key)).collect();
    let serial = 32.0 / t0.elapsed().as_secs_f64();
    let t1 = std::time::Instant::now();
    let team: Vec<Ciphertext> = pairs.par_iter().map(|(a, b)| gates.nand(a, b, &cloud_key)).collect();
    let together = 512.0 / t1.elapsed().as_secs_f64();
    for (i, (out, &(a, b))) in team.iter().zip(plains.iter()).enumerate() {
        assert_eq!(out.decrypt_bool(&key.key_lv0), !(a & b), "gate {}", i);
    }
    for (out, single) in team.iter().zip(one_by_one.iter()) {
        assert_eq!(out.p, single.p); // merged or alone: the same bits
    }
    for (out, (a, b)) in team.iter().zip(pairs.iter()).take(16) {
        assert_eq!(out.p, cpu.nand(a, b, &cloud_key).p);
    }
    println!("one by one {:.0} gates/s, a Rayon team {:.0} gates/s", serial, together);
    assert!(rayon::current_num_threads() < 4 || together > 2.0 * serial);
}

/// vanilla.rs:127-131
#[test]
fn test_bootstrap_trait() {
    let bootstrap: Box<dyn Bootstrap> = Box::new(HipBootstrap::new());
    assert_eq!(bootstrap.name(), "hip-gfx950");
}

/// trgsw.rs:507-529 through the batch entry point, and trgsw::batch_blind_rotate word for word
#[test]
fn test_batch_blind_rotate() {
    let mut rng = rand::thread_rng();
    let (key, cloud_key) = keys();
    let plains: Vec<bool> = (0..10).map(|_| rng.gen::<bool>()).collect();
    let tlwes: Vec<tlwe::TLWELv0> = plains.iter().map(|&p| enc(p, &key)).collect();
    let gpu = gates_hip::batch_blind_rotate_hip(&tlwes, &cloud_key);
    let cpu: Vec<trlwe::TRLWELv1> = tlwes.iter().map(|t| trgsw::blind_rotate(t, &cloud_key)).collect();
    for ((g, c), &plain) in gpu.iter().zip(cpu.iter()).zip(plains.iter()) {
        assert_eq!(trlwe::sample_extract_index(g, 0).decrypt_bool(&key.key_lv1), plain);
        assert_eq!((g.a, g.b), (c.a, c.b));
    }
}

/// gates.rs:352-547: each batch function against its CPU namesake, word for word and in input order; 600 pairs are more
/// than the 256 one GPU takes before the pool cuts a batch (so a multi-GPU engine shards this one)
#[test]
fn test_batch_gates_equal_the_cpu_functions() {
    type BatchFn = fn(&[(Ciphertext, Ciphertext)], &CloudKey) -> Vec<Ciphertext>;
    let mut rng = rand::thread_rng();
    let (key, cloud_key) = keys();
    let plains: Vec<(bool, bool)> = (0..600).map(|_| (rng.gen::<bool>(), rng.gen::<bool>())).collect();
    let inputs: Vec<(Ciphertext, Ciphertext)> = plains.iter().map(|&(a, b)| (enc(a, &key), enc(b, &key))).collect();
    // the CPU side: the reference's own `_with_railgun` bodies (gates.rs:357-547), which the patch leaves untouched
    // (non-capturing closures, so that they coerce to `fn` pointers)
    let cases: [(BatchFn, BatchFn, fn(bool, bool) -> bool); 6] = [
        (gates_hip::batch_nand_hip, |i, k| gates::batch_nand_with_railgun(i, k, default_railgun()), |a, b| !(a & b)),
        (gates_hip::batch_and_hip, |i, k| gates::batch_and_with_railgun(i, k, default_railgun()), |a, b| a & b),
        (gates_hip::batch_or_hip, |i, k| gates::batch_or_with_railgun(i, k, default_railgun()), |a, b| a | b),
        (gates_hip::batch_xor_hip, |i, k| gates::batch_xor_with_railgun(i, k, default_railgun()), |a, b| a ^ b),
        (gates_hip::batch_nor_hip, |i, k| gates::batch_nor_with_railgun(i, k, default_railgun()), |a, b| !(a | b)),
        (gates_hip::batch_xnor_hip, |i, k| gates::batch_xnor_with_railgun(i, k, default_railgun()), |a, b| a ^ b),   // gates.rs:575: the reference's xnor is xor
    ];
    for (gpu_fn, cpu_fn, truth) in cases {
        let gpu = gpu_fn(&inputs, &cloud_key);
        assert_eq!(gpu.len(), inputs.len());
        for (g, &(a, b)) in gpu.iter().zip(plains.iter()) {
            assert_eq!(g.decrypt_bool(&key.key_lv0), truth(a, b));
        }
        let cpu = cpu_fn(&inputs[..24], &cloud_key);   // (the CPU path takes ~60 ms per gate per core)
        for (g, c) in gpu.iter().zip(cpu.iter()) {
            assert_eq!(g.p, c.p);
        }
    }
    // the crate's own entry points (gates::batch_*, trgsw::batch_blind_rotate), which rust/patches/ route here
    let via_crate = gates::batch_nand(&inputs[..40], &cloud_key);
    let direct = gates_hip::batch_nand_hip(&inputs[..40], &cloud_key);
    assert!(via_crate.iter().zip(direct.iter()).all(|(x, y)| x.p == y.p));
    assert_eq!(default_engine().data_transport(), "none");   // host-pointer calls: nothing moved device to device
}

#[cfg(feature = "lut-bootstrap")]
mod lut {
    use super::*;
    use rs_tfhe::bootstrap::hip::HipLutBootstrap;
    use rs_tfhe::bootstrap::lut::LutBootstrap;
    use rs_tfhe::lut::Generator;

    fn enc_msg(plain: bool, key: &SecretKey) -> Ciphertext {
        Ciphertext::encrypt_lwe_message(plain as usize, 2, params::SECURITY_128_BIT.tlwe_lv0.alpha, &key.key_lv0)
    }

    /// lut.rs:136-139, :274-279
    #[test]
    fn test_lut_bootstrap_creation() {
        assert_eq!(HipLutBootstrap::new().name(), "lut-hip-gfx950");
        let bootstrap: Box<dyn Bootstrap> = Box::new(HipLutBootstrap::new());
        assert_eq!(bootstrap.name(), "lut-hip-gfx950");
    }

    /// lut.rs:142-235: identity, NOT and constant functions at message modulus 2, each against LutBootstrap word for word
    #[test]
    fn test_identity_not_and_constant_functions() {
        let mut rng = rand::thread_rng();
        let (key, cloud_key) = keys();
        let (gpu, cpu) = (HipLutBootstrap::new(), LutBootstrap::new());
        let fns: [(fn(usize) -> usize, fn(bool) -> bool); 3] = [(|x| x, |p| p), (|x| 1 - x, |p| !p), (|_| 1, |_| true)];
        for (f, truth) in fns {
            for _ in 0..5 {
                let plain = rng.gen::<bool>();
                let encrypted = enc_msg(plain, &key);
                let bootstrapped = gpu.bootstrap_func(&encrypted, f, 2, &cloud_key);
                assert_eq!(bootstrapped.decrypt_lwe_message(2, &key.key_lv0) != 0, truth(plain));
                assert_eq!(bootstrapped.p, cpu.bootstrap_func(&encrypted, f, 2, &cloud_key).p);
            }
        }
    }

    /// lut.rs:238-271
    #[test]
    fn test_lut_reuse() {
        let mut rng = rand::thread_rng();
        let (key, cloud_key) = keys();
        let gpu = HipLutBootstrap::new();
        let lut = Generator::new(2).generate_lookup_table(|x: usize| 1 - x);
        let plains: Vec<bool> = (0..5).map(|_| rng.gen::<bool>()).collect();
        let cts: Vec<Ciphertext> = plains.iter().map(|&p| enc_msg(p, &key)).collect();
        for (ct, &plain) in cts.iter().zip(plains.iter()) {
            let bootstrapped = gpu.bootstrap_lut(ct, &lut, &cloud_key);
            assert_eq!(bootstrapped.decrypt_lwe_message(2, &key.key_lv0) != 0, !plain, "LUT reuse failed");
        }
        // one table, the whole batch in one call
        let batch = gpu.batch_bootstrap_lut(&cts, &lut, &cloud_key);
        for (b, ct) in batch.iter().zip(cts.iter()) {
            assert_eq!(b.p, gpu.bootstrap_lut(ct, &lut, &cloud_key).p);
        }
    }
}
