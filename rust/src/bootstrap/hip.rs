// rust/src/bootstrap/hip.rs -- the crate-side binding of libtfhe_hip.so for thedonutfactory/rs-tfhe.
//
// Copy to `src/bootstrap/hip.rs` of the crate and add `#[cfg(feature = "hip")] pub mod hip;` to `src/bootstrap/mod.rs`.
// UNCOMPILED: the image this repository is built in has no Rust toolchain.  What holds it to the C ABI meanwhile is
// tests/test_binding_lint.py (every `extern "C"` declaration below against include/tfhe_hip.h: arity, pointer depth,
// constness, integer widths, return type, and the symbol exported by the built library); the same API surface is
// compiled and run on the GPU through the C++ mirror include/rs_tfhe_hip.hpp (tests/cpp/test_mirror.cpp).
// Interfaces bound: src/bootstrap/mod.rs:23-43 (trait Bootstrap: Send + Sync), src/gates.rs:43-45, src/trgsw.rs:53-55;
// FFI precedent in the reference itself: src/fft/spqlios/spqlios_fft.rs:23-34.
//! MI355X bootstrap strategies: blind rotate + sample extract + key switch on the GPU(s).
use crate::bootstrap::Bootstrap;
use crate::key::CloudKey;
#[cfg(feature = "lut-bootstrap")]
use crate::lut::LookupTable;
use crate::{params, trlwe};
use crate::utils::Ciphertext;
use std::cell::RefCell;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::Mutex;

#[repr(C)]
struct TfheHipParams { n: i32, l: i32, bgbit: i32, basebit: i32, t: i32 }
#[repr(C)]
pub struct TfheHipPool { _private: [u8; 0] }

extern "C" {   // include/tfhe_hip.h, the tfhe_hip_pool_* family: one handle, 1..64 devices
    fn tfhe_hip_device_count() -> c_int;
    fn tfhe_hip_pool_create(p: *const TfheHipParams, devices: *const c_int, ndev: c_int,
                            out: *mut *mut TfheHipPool) -> c_int;
    fn tfhe_hip_pool_key_create(pool: *mut TfheHipPool, key_view: *mut *mut TfheHipPool) -> c_int;  // another resident key
    fn tfhe_hip_pool_destroy(pool: *mut TfheHipPool);   // a pool, or a key view of one (views first)
    fn tfhe_hip_pool_last_error(pool: *const TfheHipPool) -> *const c_char;
    fn tfhe_hip_pool_load_cloud_key(pool: *mut TfheHipPool, bsk: *const f64, ksk: *const u32,
                                    decomp_offset: u32, testvec: *const u32) -> c_int;
    fn tfhe_hip_pool_gen_cloud_key_with_key(pool: *mut TfheHipPool, key_lv0: *const u32, key_lv1: *const u32,
                                            alpha_ksk: f64, alpha_bsk: f64, rng_key: *const u8) -> c_int;
    fn tfhe_hip_pool_export_cloud_key(pool: *mut TfheHipPool, member: c_int, bsk: *mut f64, ksk: *mut u32,
                                      decomp_offset: *mut u32, testvec: *mut u32) -> c_int;
    fn tfhe_hip_pool_batch_gate(pool: *mut TfheHipPool, gate: c_int, a: *const u32, b: *const u32,
                                out: *mut u32, count: usize) -> c_int;
    fn tfhe_hip_pool_batch_gates_mixed(pool: *mut TfheHipPool, gates: *const u8, a: *const u32, b: *const u32,
                                       out: *mut u32, count: usize) -> c_int;
    fn tfhe_hip_pool_batch_bootstrap(pool: *mut TfheHipPool, input: *const u32, testvec: *const u32,
                                     per_ct: c_int, keyswitch: c_int, out: *mut u32, count: usize) -> c_int;
    fn tfhe_hip_pool_batch_blind_rotate(pool: *mut TfheHipPool, input: *const u32, testvec: *const u32,
                                        out_trlwe: *mut u32, count: usize) -> c_int;
    fn tfhe_hip_pool_batch_mux(pool: *mut TfheHipPool, naive: c_int, a: *const u32, b: *const u32,
                               c: *const u32, out: *mut u32, count: usize) -> c_int;
    // a batch RESIDENT on one member's GPU: device pointers, shards travel by grouped RCCL send / receive (or peer copies)
    fn tfhe_hip_pool_batch_gate_dev(pool: *mut TfheHipPool, home_member: c_int, gate: c_int, a: *const u32,
                                    b: *const u32, out: *mut u32, count: usize, stream: *mut c_void) -> c_int;
    fn tfhe_hip_pool_batch_gates_mixed_dev(pool: *mut TfheHipPool, home_member: c_int, gates: *const u8, a: *const u32,
                                           b: *const u32, out: *mut u32, count: usize, stream: *mut c_void) -> c_int;
    fn tfhe_hip_pool_batch_gates_mixed_nks_dev(pool: *mut TfheHipPool, home_member: c_int, gates: *const u8,
                                               a: *const u32, b: *const u32, out: *mut u32, count: usize,
                                               stream: *mut c_void) -> c_int;
    fn tfhe_hip_pool_batch_bootstrap_dev(pool: *mut TfheHipPool, home_member: c_int, input: *const u32,
                                         testvec: *const u32, per_ct: c_int, keyswitch: c_int, out: *mut u32,
                                         count: usize, stream: *mut c_void) -> c_int;
    fn tfhe_hip_pool_batch_lincomb_bootstrap_dev(pool: *mut TfheHipPool, home_member: c_int, ca: u32, a: *const u32,
                                                 cb: u32, b: *const u32, cconst: u32, testvec: *const u32,
                                                 per_ct: c_int, keyswitch: c_int, out: *mut u32, count: usize,
                                                 stream: *mut c_void) -> c_int;
    fn tfhe_hip_pool_batch_mux_dev(pool: *mut TfheHipPool, home_member: c_int, naive: c_int, a: *const u32,
                                   b: *const u32, c: *const u32, out: *mut u32, count: usize,
                                   stream: *mut c_void) -> c_int;
    fn tfhe_hip_pool_synchronize(pool: *mut TfheHipPool) -> c_int;
    fn tfhe_hip_pool_data_transport(pool: *const TfheHipPool) -> *const c_char;
    fn tfhe_hip_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;   // pinned host memory
    fn tfhe_hip_host_free(p: *mut c_void);
}

pub const NAND: c_int = 0; pub const OR: c_int = 1; pub const AND: c_int = 2; pub const XOR: c_int = 3;
pub const XNOR: c_int = 4; pub const NOR: c_int = 5; pub const ANDNY: c_int = 6; pub const ANDYN: c_int = 7;
pub const ORNY: c_int = 8; pub const ORYN: c_int = 9; pub const COPY: c_int = 10;

const W: usize = params::tlwe_lv0::N + 1;     // words per TLWELv0
const N: usize = params::trgsw_lv1::N;        // 1024
const MAX_RESIDENT_KEYS: usize = 4;           // key views kept per pool (172 MB + 104 MB of byte planes each)

/// Grow-only pinned buffer (tfhe_hip_host_alloc): operands flattened into it are read / written in place by the GPU.
struct Pinned { p: *mut u32, words: usize }
impl Pinned {
    const fn new() -> Self { Pinned { p: std::ptr::null_mut(), words: 0 } }
    fn get(&mut self, words: usize) -> &mut [u32] {
        if words > self.words {
            unsafe { if !self.p.is_null() { tfhe_hip_host_free(self.p as *mut c_void); } }
            let mut q: *mut c_void = std::ptr::null_mut();
            let want = words + words / 4;
            assert_eq!(unsafe { tfhe_hip_host_alloc(want * 4, &mut q) }, 0, "tfhe_hip_host_alloc failed");
            self.p = q as *mut u32; self.words = want;
        }
        unsafe { std::slice::from_raw_parts_mut(self.p, words) }
    }
}
impl Drop for Pinned { fn drop(&mut self) { unsafe { if !self.p.is_null() { tfhe_hip_host_free(self.p as *mut c_void); } } } }
thread_local! {   // one arena per calling thread: a, b, c operands and the result
    static ARENA: RefCell<[Pinned; 4]> = RefCell::new([Pinned::new(), Pinned::new(), Pinned::new(), Pinned::new()]);
}

/// A resident key: (address, content sample) of the CloudKey it holds + its key view of the pool.
struct KeyView { addr: usize, fp: u64, view: *mut TfheHipPool, last_use: u64, users: usize }

/// Owns the C pool (one context per device) and the key views on it.
pub struct HipEngine { pool: *mut TfheHipPool, views: Mutex<(Vec<KeyView>, u64)> }
unsafe impl Send for HipEngine {}   // the library merges concurrent small calls into shared launches and serialises the rest per context; `views` is behind its Mutex
unsafe impl Sync for HipEngine {}

impl HipEngine {
    /// `devices`: HIP device indices, e.g. `&[0]` or `&[0, 1, 2, 3, 4, 5, 6, 7]` (all GPUs of a node).  Batch
    /// calls split their slice contiguously over the devices, exactly as `par_iter().map().collect()`
    /// (src/parallel/rayon_impl.rs:40-47) keeps input order.
    pub fn new(devices: &[i32]) -> Self {
        let p = TfheHipParams {
            n: params::tlwe_lv0::N as i32, l: params::trgsw_lv1::L as i32,
            bgbit: params::trgsw_lv1::BGBIT as i32, basebit: params::trgsw_lv1::BASEBIT as i32,
            t: params::trgsw_lv1::IKS_T as i32,
        };
        let mut pool = std::ptr::null_mut();
        let rc = unsafe { tfhe_hip_pool_create(&p, devices.as_ptr(), devices.len() as c_int, &mut pool) };
        assert_eq!(rc, 0, "tfhe_hip_pool_create failed");   // the reference has no Result on this path
        HipEngine { pool, views: Mutex::new((Vec::new(), 0)) }
    }

    fn check(h: *mut TfheHipPool, rc: c_int) {
        if rc != 0 {
            let msg = unsafe { std::ffi::CStr::from_ptr(tfhe_hip_pool_last_error(h)) };
            panic!("tfhe_hip: {}", msg.to_string_lossy());
        }
    }

    /// a content sample of both keys + sizes (FNV-style mix), as the C++ mirror takes one
    fn fingerprint(ck: &CloudKey) -> u64 {
        let mut h: u64 = 0x9E37_79B9_7F4A_7C15 ^ ck.decomposition_offset as u64;
        let mut mix = |v: u64| { h = (h ^ v).wrapping_mul(0x0000_0100_0000_01B3); };
        let nk = ck.key_switching_key.len();
        for i in 0..64 { let t = &ck.key_switching_key[(nk - 1) - (nk - 1) * i / 64]; mix(t.p[i % W] as u64); }
        let nb = ck.bootstrapping_key.len();
        for i in 0..64 { let row = &ck.bootstrapping_key[(nb - 1) * i / 64].rows()[0]; mix(row.b[i].to_bits()); }
        mix(nk as u64); mix(nb as u64);
        h
    }

    /// CloudKey is not #[repr(C)] and TRGSWLv1FFT's field is private (src/trgsw.rs:53-55): marshal
    /// field by field into the flat layouts of tfhe_hip.h.  Needs one accessor in trgsw.rs:
    ///     impl TRGSWLv1FFT { pub fn rows(&self) -> &[trlwe::TRLWELv1FFT; L * 2] { &self.trlwe_fft } }
    fn upload(view: *mut TfheHipPool, ck: &CloudKey) {
        let mut bsk: Vec<f64> = Vec::with_capacity(ck.bootstrapping_key.len() * 2 * params::trgsw_lv1::L * 2 * N);
        for trgsw in ck.bootstrapping_key.iter() {
            for row in trgsw.rows().iter() { bsk.extend_from_slice(&row.a); bsk.extend_from_slice(&row.b); }
        }
        let mut ksk: Vec<u32> = Vec::with_capacity(ck.key_switching_key.len() * W);
        for t in ck.key_switching_key.iter() { ksk.extend_from_slice(&t.p); }
        let mut tv: Vec<u32> = Vec::with_capacity(2 * N);
        tv.extend_from_slice(&ck.blind_rotate_testvec.a);
        tv.extend_from_slice(&ck.blind_rotate_testvec.b);
        // uploaded to the first device once, replicated to the others device to device
        Self::check(view, unsafe { tfhe_hip_pool_load_cloud_key(view, bsk.as_ptr(), ksk.as_ptr(),
                                                                ck.decomposition_offset, tv.as_ptr()) });
    }

    /// Run `call(view)` under the key view of `ck`: found by (address, fingerprint), else created (dropping the least
    /// recently used IDLE view beyond MAX_RESIDENT_KEYS) and loaded.  The registry lock covers lookup / creation /
    /// upload only; the batch call itself runs outside it (the library serialises per context), so threads with
    /// different keys do not queue behind one another's 330 ms batches at this level.
    fn with_key<R>(&self, ck: &CloudKey, call: impl FnOnce(*mut TfheHipPool) -> (c_int, R)) -> R {
        let (addr, fp) = (ck as *const CloudKey as usize, Self::fingerprint(ck));
        let view = {
            let mut g = self.views.lock().unwrap();
            g.1 += 1;
            let tick = g.1;
            let views = &mut g.0;
            let idx = match views.iter().position(|v| v.addr == addr && v.fp == fp) {
                Some(i) => i,
                None => {
                    while views.len() >= MAX_RESIDENT_KEYS {
                        match views.iter().enumerate().filter(|(_, v)| v.users == 0).min_by_key(|(_, v)| v.last_use) {
                            Some((i, _)) => { unsafe { tfhe_hip_pool_destroy(views[i].view) }; views.remove(i); }
                            None => break,   // every view is in use: exceed the cap for now
                        }
                    }
                    let mut v = std::ptr::null_mut();
                    assert_eq!(unsafe { tfhe_hip_pool_key_create(self.pool, &mut v) }, 0, "tfhe_hip_pool_key_create failed");
                    Self::upload(v, ck);   // under the registry lock: a second thread with the same new key waits here
                    views.push(KeyView { addr, fp, view: v, last_use: 0, users: 0 });
                    views.len() - 1
                }
            };
            views[idx].users += 1;
            views[idx].last_use = tick;
            views[idx].view
        };
        let (rc, r) = call(view);
        { let mut g = self.views.lock().unwrap(); if let Some(v) = g.0.iter_mut().find(|v| v.view == view) { v.users -= 1; } }
        Self::check(view, rc);
        r
    }

    /// flatten ciphertexts into pinned arena slot `slot`; the slice stays valid until the thread's next call
    fn flatten_into<'a>(arena: &'a mut [Pinned; 4], slot: usize, cts: impl Iterator<Item = impl std::borrow::Borrow<Ciphertext>>, n: usize) -> *const u32 {
        let buf = arena[slot].get(n * W);
        for (i, c) in cts.enumerate() { buf[i * W..(i + 1) * W].copy_from_slice(&c.borrow().p); }
        buf.as_ptr()
    }
    fn unflatten(flat: &[u32]) -> Vec<Ciphertext> {
        flat.chunks_exact(W).map(|c| { let mut t = Ciphertext::new(); t.p.copy_from_slice(c); t }).collect()
    }

    /// gates::batch_* (src/gates.rs:352-547): prep + blind rotate + extract + key switch, all devices
    pub fn batch_gate(&self, gate: c_int, inputs: &[(Ciphertext, Ciphertext)], ck: &CloudKey) -> Vec<Ciphertext> {
        ARENA.with(|ar| {
            let ar = &mut *ar.borrow_mut();
            let a = Self::flatten_into(ar, 0, inputs.iter().map(|p| &p.0), inputs.len());
            let b = Self::flatten_into(ar, 1, inputs.iter().map(|p| &p.1), inputs.len());
            let out = ar[3].get(inputs.len() * W).as_mut_ptr();
            self.with_key(ck, |v| (unsafe { tfhe_hip_pool_batch_gate(v, gate, a, b, out, inputs.len()) }, ()));
            Self::unflatten(unsafe { std::slice::from_raw_parts(out, inputs.len() * W) })
        })
    }

    /// one launch per circuit level, one gate code per ciphertext (examples/add_two_numbers.rs)
    pub fn batch_gates_mixed(&self, gates: &[u8], inputs: &[(Ciphertext, Ciphertext)], ck: &CloudKey) -> Vec<Ciphertext> {
        assert_eq!(gates.len(), inputs.len());
        ARENA.with(|ar| {
            let ar = &mut *ar.borrow_mut();
            let a = Self::flatten_into(ar, 0, inputs.iter().map(|p| &p.0), inputs.len());
            let b = Self::flatten_into(ar, 1, inputs.iter().map(|p| &p.1), inputs.len());
            let out = ar[3].get(inputs.len() * W).as_mut_ptr();
            self.with_key(ck, |v| (unsafe { tfhe_hip_pool_batch_gates_mixed(v, gates.as_ptr(), a, b, out, inputs.len()) }, ()));
            Self::unflatten(unsafe { std::slice::from_raw_parts(out, inputs.len() * W) })
        })
    }

    /// Bootstrap::bootstrap / bootstrap_without_key_switch / LutBootstrap::bootstrap_lut over a batch
    pub fn batch_bootstrap(&self, cts: &[Ciphertext], testvec: Option<&trlwe::TRLWELv1>, keyswitch: bool, ck: &CloudKey) -> Vec<Ciphertext> {
        let tv: Option<Vec<u32>> = testvec.map(|t| { let mut v = Vec::with_capacity(2 * N); v.extend_from_slice(&t.a); v.extend_from_slice(&t.b); v });
        let tvp = tv.as_ref().map_or(std::ptr::null(), |v| v.as_ptr());
        ARENA.with(|ar| {
            let ar = &mut *ar.borrow_mut();
            let a = Self::flatten_into(ar, 0, cts.iter(), cts.len());
            let out = ar[3].get(cts.len() * W).as_mut_ptr();
            self.with_key(ck, |v| (unsafe { tfhe_hip_pool_batch_bootstrap(v, a, tvp, 0, keyswitch as c_int, out, cts.len()) }, ()));
            Self::unflatten(unsafe { std::slice::from_raw_parts(out, cts.len() * W) })
        })
    }

    /// trgsw::batch_blind_rotate (src/trgsw.rs:289-294)
    pub fn batch_blind_rotate(&self, srcs: &[Ciphertext], ck: &CloudKey) -> Vec<trlwe::TRLWELv1> {
        ARENA.with(|ar| {
            let ar = &mut *ar.borrow_mut();
            let a = Self::flatten_into(ar, 0, srcs.iter(), srcs.len());
            let mut out = vec![0u32; srcs.len() * 2 * N];   // 8 KiB per sample: staged by the library
            self.with_key(ck, |v| (unsafe { tfhe_hip_pool_batch_blind_rotate(v, a, std::ptr::null(), out.as_mut_ptr(), srcs.len()) }, ()));
            out.chunks_exact(2 * N).map(|c| { let mut t = trlwe::TRLWELv1::new(); t.a.copy_from_slice(&c[..N]); t.b.copy_from_slice(&c[N..]); t }).collect()
        })
    }

    /// Gates::mux (the reference's formula, gates.rs:157-183) / Gates::mux_naive (:189-199) over a batch
    pub fn batch_mux(&self, naive: bool, abc: &[(Ciphertext, Ciphertext, Ciphertext)], ck: &CloudKey) -> Vec<Ciphertext> {
        ARENA.with(|ar| {
            let ar = &mut *ar.borrow_mut();
            let a = Self::flatten_into(ar, 0, abc.iter().map(|t| &t.0), abc.len());
            let b = Self::flatten_into(ar, 1, abc.iter().map(|t| &t.1), abc.len());
            let c = Self::flatten_into(ar, 2, abc.iter().map(|t| &t.2), abc.len());
            let out = ar[3].get(abc.len() * W).as_mut_ptr();
            self.with_key(ck, |v| (unsafe { tfhe_hip_pool_batch_mux(v, naive as c_int, a, b, c, out, abc.len()) }, ()));
            Self::unflatten(unsafe { std::slice::from_raw_parts(out, abc.len() * W) })
        })
    }

    /// The same map for a batch that is ALREADY RESIDENT on member `home`'s GPU (the levels of a circuit, the output of a
    /// previous call): `a`, `b`, `out` are device pointers to `count` rows of n + 1 words on that GPU, `stream` a
    /// hipStream_t of it (null = the member's own).  The library cuts the batch over the members, moves the shards by
    /// grouped RCCL send / receive over its persistent communicator (peer copies when the pool repeats a device),
    /// bootstraps them in parallel and orders the gathered result into `stream`; the call only enqueues.
    /// Safety: the pointers must stay valid until the work has run (`synchronize`).
    pub unsafe fn batch_gate_dev(&self, home: usize, gate: c_int, a: *const u32, b: *const u32, out: *mut u32,
                                 count: usize, stream: *mut c_void, ck: &CloudKey) {
        self.with_key(ck, |v| (tfhe_hip_pool_batch_gate_dev(v, home as c_int, gate, a, b, out, count, stream), ()));
    }
    /// One circuit level on the device: per-ciphertext gate codes (device pointer), with or without the key switch
    /// (`keyswitch = false` is the first level of Gates::mux, gates.rs:165-177).
    pub unsafe fn batch_gates_mixed_dev(&self, home: usize, gates: *const u8, a: *const u32, b: *const u32, out: *mut u32,
                                        count: usize, keyswitch: bool, stream: *mut c_void, ck: &CloudKey) {
        self.with_key(ck, |v| (if keyswitch { tfhe_hip_pool_batch_gates_mixed_dev(v, home as c_int, gates, a, b, out, count, stream) }
                               else { tfhe_hip_pool_batch_gates_mixed_nks_dev(v, home as c_int, gates, a, b, out, count, stream) }, ()));
    }
    /// Drain what the device-resident calls enqueued on the members' own streams.
    pub fn synchronize(&self) { Self::check(self.pool, unsafe { tfhe_hip_pool_synchronize(self.pool) }); }
    /// "rccl" / "peer-copy" / "none": how the last device-resident call moved its shards.
    pub fn data_transport(&self) -> &'static str {
        unsafe { std::ffi::CStr::from_ptr(tfhe_hip_pool_data_transport(self.pool)) }.to_str().unwrap_or("?")
    }

    /// CloudKey::new(&secret_key) on the GPU (src/key.rs:59-66): replaces the sequential key-switching-key loop
    /// of key.rs:107-119 (the slowest user-visible step of the reference) and the 172 MB upload.  Masks and noise
    /// are a ChaCha20 stream under 32 bytes drawn from the crate's own CSPRNG -- `rand::rngs::OsRng`, the source
    /// thread_rng is seeded from -- never from a fixed seed: whoever can regenerate the noise reads the secret key
    /// off the published key rows.  The key is generated in a FRESH key view (no other call can see or disturb it)
    /// and read back in the reference layouts; the caller builds its CloudKey from the flat arrays (the inverse of
    /// `upload`) and the view is dropped -- or kept registered under the new CloudKey's address to skip the first upload.
    pub fn gen_cloud_key(&self, sk: &crate::key::SecretKey) -> (Vec<f64>, Vec<u32>, u32, Vec<u32>) {
        use rand::RngCore;
        let mut rng_key = [0u8; 32];
        rand::rngs::OsRng.fill_bytes(&mut rng_key);
        let mut view = std::ptr::null_mut();
        assert_eq!(unsafe { tfhe_hip_pool_key_create(self.pool, &mut view) }, 0, "tfhe_hip_pool_key_create failed");
        Self::check(view, unsafe { tfhe_hip_pool_gen_cloud_key_with_key(view, sk.key_lv0.as_ptr(), sk.key_lv1.as_ptr(),
                                                                        params::tlwe_lv0::ALPHA, params::tlwe_lv1::ALPHA,
                                                                        rng_key.as_ptr()) });
        rng_key.iter_mut().for_each(|b| *b = 0);
        let n = params::tlwe_lv0::N;
        let mut bsk = vec![0f64; n * 2 * params::trgsw_lv1::L * 2 * N];
        let mut ksk = vec![0u32; N * params::trgsw_lv1::IKS_T * (1 << params::trgsw_lv1::BASEBIT) * W];
        let (mut off, mut tv) = (0u32, vec![0u32; 2 * N]);
        Self::check(view, unsafe { tfhe_hip_pool_export_cloud_key(view, 0, bsk.as_mut_ptr(), ksk.as_mut_ptr(), &mut off, tv.as_mut_ptr()) });
        unsafe { tfhe_hip_pool_destroy(view) };
        (bsk, ksk, off, tv)
    }
}
/// The process-wide engine behind `default_bootstrap()` and the `gates::batch_*` functions when the crate is built with
/// `--features hip`: created on first use over every GPU of the node -- the stand-in for `default_railgun()`'s "one
/// worker per logical CPU" (src/parallel/mod.rs:79-97, rayon_impl.rs:15-27) -- or over the devices listed in
/// `TFHE_HIP_DEVICES` ("0", "0,1,2,3", ...).  It lives for the rest of the process (as Rayon's global pool does).
pub fn default_engine() -> std::sync::Arc<HipEngine> {
    static ENGINE: std::sync::OnceLock<std::sync::Arc<HipEngine>> = std::sync::OnceLock::new();
    ENGINE.get_or_init(|| {
        let devices: Vec<i32> = match std::env::var("TFHE_HIP_DEVICES") {
            Ok(list) => list.split(',').map(|d| d.trim().parse().expect("TFHE_HIP_DEVICES: comma-separated device indices")).collect(),
            Err(_) => (0..unsafe { tfhe_hip_device_count() }).collect(),
        };
        assert!(!devices.is_empty(), "tfhe_hip: no GPU visible to this process");
        std::sync::Arc::new(HipEngine::new(&devices))
    }).clone()
}

impl Drop for HipEngine {
    fn drop(&mut self) {
        for v in self.views.lock().unwrap().0.drain(..) { unsafe { tfhe_hip_pool_destroy(v.view) } }   // views before their pool
        unsafe { tfhe_hip_pool_destroy(self.pool) }
    }
}

/// The GPU stand-in for VanillaBootstrap (src/bootstrap/vanilla.rs:22-69); single calls are count = 1 batches
/// (they take the eight-waves-per-ciphertext latency kernel: 2.2 ms per gate).
pub struct HipBootstrap { engine: std::sync::Arc<HipEngine> }
impl HipBootstrap {
    /// `HipBootstrap::new()` mirrors `VanillaBootstrap::new()` (vanilla.rs:27-37): the process-wide engine.
    pub fn new() -> Self { HipBootstrap { engine: default_engine() } }
    pub fn with_engine(engine: std::sync::Arc<HipEngine>) -> Self { HipBootstrap { engine } }
    pub fn engine(&self) -> &std::sync::Arc<HipEngine> { &self.engine }
}
impl Default for HipBootstrap { fn default() -> Self { Self::new() } }

impl Bootstrap for HipBootstrap {
    fn bootstrap(&self, ctxt: &Ciphertext, cloud_key: &CloudKey) -> Ciphertext {
        self.engine.batch_bootstrap(std::slice::from_ref(ctxt), None, true, cloud_key).pop().unwrap()
    }
    fn bootstrap_without_key_switch(&self, ctxt: &Ciphertext, cloud_key: &CloudKey) -> Ciphertext {
        self.engine.batch_bootstrap(std::slice::from_ref(ctxt), None, false, cloud_key).pop().unwrap()
    }
    fn name(&self) -> &str { "hip-gfx950" }
}

/// src/bootstrap/lut.rs:24-126 on the GPU: the LUT's polynomial is the test vector of the blind rotation.
/// (Everything from here on needs the crate's `lut-bootstrap` feature, as `bootstrap::lut` itself does.)
#[cfg(feature = "lut-bootstrap")]
pub struct HipLutBootstrap { engine: std::sync::Arc<HipEngine> }
#[cfg(feature = "lut-bootstrap")]
impl HipLutBootstrap {
    /// mirrors `LutBootstrap::new()` (lut.rs:29-35): the process-wide engine
    pub fn new() -> Self { HipLutBootstrap { engine: default_engine() } }
    pub fn with_engine(engine: std::sync::Arc<HipEngine>) -> Self { HipLutBootstrap { engine } }
    /// lut.rs:49-65
    pub fn bootstrap_func<F: Fn(usize) -> usize>(&self, ct_in: &Ciphertext, f: F, message_modulus: usize, cloud_key: &CloudKey) -> Ciphertext {
        let lut = crate::lut::Generator::new(message_modulus).generate_lookup_table(f);
        self.bootstrap_lut(ct_in, &lut, cloud_key)
    }
    /// lut.rs:79-99
    pub fn bootstrap_lut(&self, ct_in: &Ciphertext, lut: &LookupTable, cloud_key: &CloudKey) -> Ciphertext {
        self.engine.batch_bootstrap(std::slice::from_ref(ct_in), Some(&lut.poly), true, cloud_key).pop().unwrap()
    }
    /// the batched form the GPU is meant to be fed with: one LUT, many ciphertexts
    pub fn batch_bootstrap_lut(&self, cts: &[Ciphertext], lut: &LookupTable, cloud_key: &CloudKey) -> Vec<Ciphertext> {
        self.engine.batch_bootstrap(cts, Some(&lut.poly), true, cloud_key)
    }
}
#[cfg(feature = "lut-bootstrap")]
impl Default for HipLutBootstrap { fn default() -> Self { Self::new() } }
#[cfg(feature = "lut-bootstrap")]
impl Bootstrap for HipLutBootstrap {
    fn bootstrap(&self, ctxt: &Ciphertext, cloud_key: &CloudKey) -> Ciphertext {     // lut.rs:108-111: identity, m = 2
        self.bootstrap_func(ctxt, |x| x, 2, cloud_key)
    }
    fn bootstrap_without_key_switch(&self, ctxt: &Ciphertext, cloud_key: &CloudKey) -> Ciphertext {   // lut.rs:113-121
        self.bootstrap(ctxt, cloud_key)
    }
    fn name(&self) -> &str { "lut-hip-gfx950" }
}
