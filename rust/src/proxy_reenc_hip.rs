// rust/src/proxy_reenc_hip.rs -- proxy_reenc::reencrypt_tlwe_lv0 (src/proxy_reenc.rs:468-510) on the GPU: copy to
// `src/proxy_reenc_hip.rs` of the crate (rust/apply.sh; module under features `proxy-reenc` + `hip`).  UNCOMPILED (no Rust toolchain in this
// image); tests/test_binding_lint.py holds the `extern "C"` block to include/tfhe_hip.h.  The same calls are compiled and
// run on the GPU in C++ (`proxy_reenc::` in include/rs_tfhe_hip.hpp, tests/cpp/test_mirror.cpp).
use crate::params;
use crate::proxy_reenc::ProxyReencryptionKey;
use crate::tlwe::TLWELv0;
use std::os::raw::{c_char, c_int};

#[repr(C)]
struct TfheHipParams { n: i32, l: i32, bgbit: i32, basebit: i32, t: i32 }
#[repr(C)]
pub struct TfheHipCtx { _private: [u8; 0] }

extern "C" {   // include/tfhe_hip.h: a context, and the two proxy re-encryption entry points
    fn tfhe_hip_ctx_create(p: *const TfheHipParams, device: c_int, out: *mut *mut TfheHipCtx) -> c_int;
    fn tfhe_hip_ctx_destroy(ctx: *mut TfheHipCtx);
    fn tfhe_hip_last_error(ctx: *const TfheHipCtx) -> *const c_char;
    fn tfhe_hip_load_reenc_key(ctx: *mut TfheHipCtx, key: *const u32) -> c_int;
    fn tfhe_hip_batch_reencrypt(ctx: *mut TfheHipCtx, input: *const u32, out: *mut u32, count: usize) -> c_int;
}

const W: usize = params::tlwe_lv0::N + 1;

/// A ProxyReencryptionKey resident on one GPU.  The context is created with THIS key's (base, t) -- the defaults of
/// new_symmetric / new_asymmetric are params::trgsw_lv1::{BASEBIT, IKS_T} (proxy_reenc.rs:271-279, :362-370) -- and
/// holds the key as a key-switching key whose coefficients n .. N-1 are never selected.
pub struct HipReencKey { ctx: *mut TfheHipCtx }
unsafe impl Send for HipReencKey {}   // the library serialises calls per context
unsafe impl Sync for HipReencKey {}

impl HipReencKey {
    pub fn new(reenc_key: &ProxyReencryptionKey, device: i32) -> Self {
        assert!(reenc_key.base.is_power_of_two(), "decomposition base must be a power of two");
        let p = TfheHipParams {
            n: params::tlwe_lv0::N as i32, l: params::trgsw_lv1::L as i32, bgbit: params::trgsw_lv1::BGBIT as i32,
            basebit: reenc_key.base.trailing_zeros() as i32, t: reenc_key.t as i32,
        };
        let mut ctx = std::ptr::null_mut();
        assert_eq!(unsafe { tfhe_hip_ctx_create(&p, device, &mut ctx) }, 0, "tfhe_hip_ctx_create failed");
        let mut flat: Vec<u32> = Vec::with_capacity(reenc_key.key_encryptions.len() * W);
        for e in reenc_key.key_encryptions.iter() { flat.extend_from_slice(&e.p); }   // index base*t*i + base*j + k
        let k = HipReencKey { ctx };
        k.check(unsafe { tfhe_hip_load_reenc_key(ctx, flat.as_ptr()) });
        k
    }
    fn check(&self, rc: c_int) {
        if rc != 0 {
            let msg = unsafe { std::ffi::CStr::from_ptr(tfhe_hip_last_error(self.ctx)) };
            panic!("tfhe_hip: {}", msg.to_string_lossy());   // the reference has no Result on this path
        }
    }
    /// reencrypt_tlwe_lv0 over a batch: one launch of the key-switch kernels for all of `cts`
    pub fn reencrypt(&self, cts: &[TLWELv0]) -> Vec<TLWELv0> {
        let mut flat: Vec<u32> = Vec::with_capacity(cts.len() * W);
        for c in cts { flat.extend_from_slice(&c.p); }
        let mut out = vec![0u32; cts.len() * W];
        self.check(unsafe { tfhe_hip_batch_reencrypt(self.ctx, flat.as_ptr(), out.as_mut_ptr(), cts.len()) });
        out.chunks_exact(W).map(|w| { let mut t = TLWELv0::new(); t.p.copy_from_slice(w); t }).collect()
    }
}
impl Drop for HipReencKey { fn drop(&mut self) { unsafe { tfhe_hip_ctx_destroy(self.ctx) } } }

/// Drop-in for `reencrypt_tlwe_lv0(&ct, &reenc_key)` once the key is resident.
pub fn reencrypt_tlwe_lv0_hip(ct_from: &TLWELv0, key: &HipReencKey) -> TLWELv0 {
    key.reencrypt(std::slice::from_ref(ct_from)).pop().unwrap()
}
