/*
 * tfhe_hip.h -- C ABI of the MI355X (gfx950) TFHE gate-bootstrapping engine.
 *
 * This is the drop-in boundary for the hot path of thedonutfactory/rs-tfhe:
 *   gate linear prep -> blind rotation -> sample extract -> identity key switch.
 * Conventions follow the reference's only existing FFI, the SPQLIOS wrapper
 * (src/fft/spqlios/spqlios-wrapper.cpp:10-41, Rust decls spqlios_fft.rs:23-34):
 * opaque handle from *_create, freed by *_destroy; caller-owned, caller-allocated
 * flat u32 / f64 buffers passed as raw pointers; no callbacks, no torch types.
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the rs-tfhe repository root).  The reference has no Result on this path
 * (failure = panic); here every function returns 0 on success or a negative
 * TFHE_HIP_E* code, with tfhe_hip_last_error() giving the text.  A binding that
 * wants the reference's infallible signatures `expect()`s the code.
 *
 * Layouts crossing the boundary (all row-major, little-endian):
 *   TLWELv0     [n+1]  u32      (src/tlwe.rs:12-14)       p[0..n]=a, p[n]=b
 *   TLWELv1     [N+1]  u32      (src/tlwe.rs:217-219)
 *   TRLWELv1    [2][N] u32      (src/trlwe.rs:11-14)      a then b
 *   TRGSWLv1FFT [2l][2][N] f64  (src/trgsw.rs:53-55, src/trlwe.rs:85-88)
 *                                each [N] = re[0..N/2] || im[0..N/2] as produced
 *                                by KlemsaProcessor::ifft (src/fft/klemsa.rs:88-117)
 *   bootstrapping_key  [n] TRGSWLv1FFT                     (src/key.rs:51-56)
 *   key_switching_key  [N][t][base][n+1] u32, index base*t*i + base*j + k
 *                                                          (src/key.rs:102-122)
 * N = 1024 in every parameter set of the reference (src/params.rs).
 *
 * Pointers are HOST pointers unless the function name ends in _dev, in which
 * case they are device pointers on the context's GPU and the call is enqueued
 * on `stream` (a hipStream_t passed as void*; NULL = the context's own non-blocking
 * stream; pass hipStreamLegacy, i.e. (hipStream_t)1, to name the default stream)
 * without synchronising.  Intermediate buffers belong to the context: when a
 * call arrives on a different stream than the previous one, the library first
 * drains the previous stream (hipStreamSynchronize), so mixing streams on one
 * context is safe but serialising; use one context per stream for overlap.
 *
 * Thread safety: a context may be used from several host threads (the
 * reference's `Bootstrap: Send + Sync`, src/bootstrap/mod.rs:23).  Concurrent SMALL
 * host-pointer calls (tfhe_hip_batch_gate / _gates_mixed[_nks] / _bootstrap /
 * _lincomb_bootstrap / _mux / _blind_rotate of up to 2 x #CUs ciphertexts, and their
 * tfhe_hip_pool_* forms) are MERGED: whatever
 * calls arrive while a launch is running share the next one (one launch per key
 * view and operation class, per-ciphertext gate codes / test vectors), so T threads
 * that each evaluate one gate get T gates per launch time, not one -- what a Rayon
 * team calling one strategy expects of its cores (src/parallel/rayon_impl.rs:40-47).
 * Results are the same bits as the unmerged call; a lone caller sees the latency of
 * a one-ciphertext call; errors stay with the thread that caused them.  See
 * tfhe_hip_set_combining.  Every other call on one context is serialised
 * internally, first come first served (a thread that issues calls back to back
 * cannot starve the others).
 */
#ifndef TFHE_HIP_H
#define TFHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFHE_HIP_N 1024

/* error codes */
#define TFHE_HIP_OK 0
#define TFHE_HIP_EINVAL (-1)  /* bad argument / unsupported parameter set */
#define TFHE_HIP_EHIP (-2)    /* a HIP runtime call failed               */
#define TFHE_HIP_ENOKEY (-3)  /* cloud key not loaded                     */
#define TFHE_HIP_ENOMEM (-4)

/* Run-time form of SecurityParams / TrgswParams (src/params.rs:53-84). */
typedef struct tfhe_hip_params {
  int32_t n;       /* tlwe_lv0.n   (<= 1279)            */
  int32_t l;       /* trgsw_lv1.l  (1..3)               */
  int32_t bgbit;   /* trgsw_lv1.bgbit (l*bgbit <= 32)   */
  int32_t basebit; /* trgsw_lv1.basebit                 */
  int32_t t;       /* trgsw_lv1.iks_t (basebit*t <= 31) */
} tfhe_hip_params;

/* Gate selector for tfhe_hip_batch_gate*: the linear prep of src/gates.rs:54-150. */
typedef enum tfhe_hip_gate {
  TFHE_HIP_NAND = 0,   /* gates.rs:54-58   -(a+b), b += 1/8  */
  TFHE_HIP_OR = 1,     /* gates.rs:62-66    a+b,   b += 1/8  */
  TFHE_HIP_AND = 2,    /* gates.rs:70-74    a+b,   b -= 1/8  */
  TFHE_HIP_XOR = 3,    /* gates.rs:78-82    a+2b,  b += 1/4  */
  TFHE_HIP_XNOR = 4,   /* gates.rs:86-90    a-2b,  b -= 1/4  */
  TFHE_HIP_NOR = 5,    /* gates.rs:94-98   -(a+b), b -= 1/8  */
  TFHE_HIP_ANDNY = 6,  /* gates.rs:102-111 -a+b,   b -= 1/8  */
  TFHE_HIP_ANDYN = 7,  /* gates.rs:115-124  a-b,   b -= 1/8  */
  TFHE_HIP_ORNY = 8,   /* gates.rs:128-137 -a+b,   b += 1/8  */
  TFHE_HIP_ORYN = 9,   /* gates.rs:141-150  a-b,   b += 1/8  */
  TFHE_HIP_COPY = 10   /* no prep: plain bootstrap of a (vanilla.rs:40-52) */
} tfhe_hip_gate;

typedef struct tfhe_hip_ctx tfhe_hip_ctx;

/* ---- lifetime ---------------------------------------------------------- */

/* Replaces: FFT_PLAN thread-local construction (src/fft/mod.rs:47-61) and the
 * SPQLIOS precedent Spqlios_new (spqlios-wrapper.cpp:10-13).
 * `device` is the HIP device ordinal this context owns (one context per GPU;
 * one process per GPU under torch.distributed). */
int tfhe_hip_ctx_create(const tfhe_hip_params *params, int device, tfhe_hip_ctx **out);

/* Replaces: Drop for SpqliosFFT (spqlios_fft.rs:84-90). NULL is a no-op. */
void tfhe_hip_ctx_destroy(tfhe_hip_ctx *ctx);

/* Text of the CALLING THREAD's last error on this context (or of its last failed create when ctx == NULL).
 * Never NULL.  The text is kept per (thread, handle): threads that share a context (`Send + Sync`,
 * src/bootstrap/mod.rs:23) each read their own failure, and the pointer stays valid until the same thread's next
 * failing call on the same handle.  A thread that has not failed on the handle reads "" (it never sees another
 * thread's text, and reading stores nothing); a thread's table of texts is bounded: entries of destroyed handles are
 * dropped once it holds 64.  tfhe_hip_pool_last_error follows the same rule. */
const char *tfhe_hip_last_error(const tfhe_hip_ctx *ctx);

/* ---- key views: several resident cloud keys on ONE context -----------------------------------------
 * Replaces: the `&CloudKey` every call of the reference names (src/bootstrap/mod.rs:23-38, src/gates.rs:43-45;
 * strategies and keys are Send + Sync).  A context holds one cloud key of its own; tfhe_hip_key_create adds
 * another resident key to it and returns a KEY VIEW: a tfhe_hip_ctx handle that EVERY entry point of this header
 * accepts in place of the context -- tfhe_hip_load_cloud_key / tfhe_hip_gen_cloud_key* / tfhe_hip_cloud_key_buffers
 * + tfhe_hip_adopt_cloud_key fill the view's key, tfhe_hip_export_cloud_key reads it, every tfhe_hip_batch_* call
 * evaluates under it -- while device, streams, scratch buffers, profiling state and the mutex are the parent's.
 * So a call names its key by the handle it passes; calls under different keys of one context may come from
 * different threads (they serialise on the parent like any two calls on one context) and cost no second set of
 * scratch buffers, streams or twiddle tables.  tfhe_hip_ctx_destroy(view) drains the parent's queued work and
 * frees only the view's key; destroy every view before its parent (a parent destroyed first stays alive, unusable
 * through its own handle, until its last view is destroyed).  tfhe_hip_last_error(view) is the parent's. */
int tfhe_hip_key_create(tfhe_hip_ctx *ctx, tfhe_hip_ctx **key_view);
/* The context a key view runs on (the handle itself for a plain context). */
tfhe_hip_ctx *tfhe_hip_key_parent(tfhe_hip_ctx *key_view);
/* 1 when the handle's own key has been loaded / generated / adopted, else 0. */
int tfhe_hip_key_is_loaded(tfhe_hip_ctx *ctx_or_view);

/* "MI355X-native HIP (gfx950)" style identification: Bootstrap::name()
 * (src/bootstrap/mod.rs:37) of the strategy this library backs. */
const char *tfhe_hip_name(void);
/* Number of GPUs this process can open (hipGetDeviceCount; 0 when there is none or the runtime cannot start): what a
 * host binding needs to build "all GPUs of the node" for tfhe_hip_pool_create -- the stand-in for Rayon's default
 * "one worker per logical CPU" (src/parallel/rayon_impl.rs:15-27).  Does not initialise any device. */
int tfhe_hip_device_count(void);

/* ---- cloud key --------------------------------------------------------- */

/* Replaces: borrowing `&CloudKey` on every call (src/key.rs:51-56).  Uploads
 * and converts the key once; host buffers may be freed on return.
 *   bsk            [n][2l][2][N] f64, reference Klemsa spectral layout
 *   ksk            [N][t][base][n+1] u32 (k = 0 slots are ignored)
 *   decomp_offset  CloudKey::decomposition_offset (key.rs:78-89)
 *   testvec        CloudKey::blind_rotate_testvec [2][N] (key.rs:91-100) */
int tfhe_hip_load_cloud_key(tfhe_hip_ctx *ctx, const double *bsk, const uint32_t *ksk,
                            uint32_t decomp_offset, const uint32_t *testvec);

/* Replaces: CloudKey::new(&secret_key) (src/key.rs:59-66) = gen_decomposition_offset (:78-89),
 * gen_testvec (:91-100), gen_key_switching_key (:102-122) and gen_bootstrapping_key (:124-156),
 * generated on the GPU straight into the context (no upload).  Needs the secret key, so it is a
 * client-side call.  key_lv0 [n], key_lv1 [N]: 0/1 words (src/key.rs:21-49).
 * alpha_ksk = tlwe_lv0.alpha (KSK_ALPHA, params.rs:468), alpha_bsk = tlwe_lv1.alpha (BSK_ALPHA,
 * :469).
 *
 * Randomness.  The reference draws every mask word and noise sample from an OS-seeded ChaCha thread_rng
 * (tlwe.rs:38, trlwe.rs:36-41).  Here they are fixed positions of a ChaCha20 keystream (RFC 8439, 20 rounds)
 * under a 256-bit generator key; the distributions are the reference's.  A published cloud key is only as
 * secret as that generator key -- whoever can regenerate the noise solves the key rows for the secret key --
 * so:
 *   tfhe_hip_gen_cloud_key_secure    the generator key comes from getrandom(2).  USE THIS ONE.
 *   tfhe_hip_gen_cloud_key_with_key  the caller supplies 32 bytes from its own CSPRNG (e.g. Rust's OsRng).
 *   tfhe_hip_gen_cloud_key           TEST / BENCHMARK ONLY: a 64-bit seed is expanded into the generator key,
 *                                    which makes the cloud key reproducible bit for bit and exactly as
 *                                    guessable as the seed.  Never publish a key generated this way.
 * The secret-key staging buffers on the device are zeroed before any of the three returns. */
int tfhe_hip_gen_cloud_key_secure(tfhe_hip_ctx *ctx, const uint32_t *key_lv0, const uint32_t *key_lv1,
                                  double alpha_ksk, double alpha_bsk);
int tfhe_hip_gen_cloud_key_with_key(tfhe_hip_ctx *ctx, const uint32_t *key_lv0, const uint32_t *key_lv1,
                                    double alpha_ksk, double alpha_bsk, const uint8_t rng_key[32]);
int tfhe_hip_gen_cloud_key(tfhe_hip_ctx *ctx, const uint32_t *key_lv0, const uint32_t *key_lv1,
                           double alpha_ksk, double alpha_bsk, uint64_t seed);

/* The loaded / generated cloud key back in the reference layouts of tfhe_hip_load_cloud_key
 * (any pointer may be NULL to skip that field).  load -> export is the identity. */
int tfhe_hip_export_cloud_key(tfhe_hip_ctx *ctx, double *bsk, uint32_t *ksk, uint32_t *decomp_offset,
                              uint32_t *testvec);

/* The context's cloud key as raw device buffers in the ENGINE layouts, for callers that
 * replicate a key device to device themselves -- one process per GPU, where the pool's hipMemcpyPeer cannot reach:
 * the owner of the key and every receiver call this (it allocates the buffers when the context has none yet),
 * the bytes travel (RCCL broadcast, hipMemcpyPeer, IPC), and each receiver then calls
 * tfhe_hip_adopt_cloud_key(ctx, decomp_offset of the owner).  The buffers stay owned by the context; while a
 * receiver's buffers are being written it must not run batches.  Any out pointer may be NULL. */
int tfhe_hip_cloud_key_buffers(tfhe_hip_ctx *ctx, void **bsk, size_t *bsk_bytes, void **ksk, size_t *ksk_bytes,
                               void **testvec, size_t *testvec_bytes, uint32_t *decomp_offset);
int tfhe_hip_adopt_cloud_key(tfhe_hip_ctx *ctx, uint32_t decomp_offset);

/* Pinned host buffers.  The host entry points below take ordinary (pageable) memory and stage it through the
 * device around the kernels: 3 x 184 MB for a 65,536-ciphertext gate batch, about 7 % of the call.  When EVERY
 * ciphertext operand of a call (inputs and output) is pinned host memory -- allocated here, or the caller's own
 * buffers registered with hipHostRegister -- the kernels read and write it in place over PCIe (each ciphertext is
 * read once in the blind rotation's prologue and written once by the key switch) and the call runs at the
 * device-resident rate.  Nothing else changes: same functions, same arguments, same results. */
int tfhe_hip_host_alloc(size_t bytes, void **out);
void tfhe_hip_host_free(void *p);

/* ---- the hot path, batched --------------------------------------------- */

/* Replaces: gates::batch_{nand,and,or,xor,nor,xnor}[_with_railgun]
 * (src/gates.rs:352-547) and, with count == 1, Gates::{nand,...,or_yn}
 * (src/gates.rs:54-150).  a, b, out: [count][n+1].  b is ignored for COPY. */
int tfhe_hip_batch_gate(tfhe_hip_ctx *ctx, int gate, const uint32_t *a, const uint32_t *b,
                        uint32_t *out, size_t count);
int tfhe_hip_batch_gate_dev(tfhe_hip_ctx *ctx, int gate, const uint32_t *a, const uint32_t *b,
                            uint32_t *out, size_t count, void *stream);

/* A batch whose ciphertexts carry different gates: gates[c] is the tfhe_hip_gate of
 * ciphertext pair c.  This is what a levelised circuit (e.g. the ripple-carry adder of
 * examples/add_two_numbers.rs) issues per level: every gate of the level in ONE launch,
 * whatever its type.  Same semantics as count calls of the Gates method
 * (src/gates.rs:54-150); b is read for every gate but COPY.  The host entry rejects codes
 * above TFHE_HIP_COPY with TFHE_HIP_EINVAL; the _dev entry cannot read device memory on the
 * host: the kernel treats such a ciphertext as COPY and raises a device-side flag that the next
 * tfhe_hip_synchronize() reports as TFHE_HIP_EINVAL. */
int tfhe_hip_batch_gates_mixed(tfhe_hip_ctx *ctx, const uint8_t *gates, const uint32_t *a,
                               const uint32_t *b, uint32_t *out, size_t count);
int tfhe_hip_batch_gates_mixed_dev(tfhe_hip_ctx *ctx, const uint8_t *gates, const uint32_t *a,
                                   const uint32_t *b, uint32_t *out, size_t count, void *stream);

/* The same per-ciphertext gates ending in bootstrap_without_key_switch (bootstrap/mod.rs:31-35): the output is
 * sample_extract_index_2 of the rotated accumulator, [count][n+1] (trlwe.rs:122-136; needs n <= N).  This is the
 * first level of Gates::mux (gates.rs:165-177: and(a, b) and and(not(a), c) without key switch) for a whole batch in
 * ONE launch -- gates = AND for the first operand pairs, ANDNY for the second; the second level (or(u1, u2),
 * gates.rs:179-182) is a tfhe_hip_batch_gates_mixed launch that can carry other gates of the same circuit level
 * beside it.  Same gate-code rules as above. */
int tfhe_hip_batch_gates_mixed_nks(tfhe_hip_ctx *ctx, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                                   uint32_t *out, size_t count);
int tfhe_hip_batch_gates_mixed_nks_dev(tfhe_hip_ctx *ctx, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                                       uint32_t *out, size_t count, void *stream);

/* Replaces: Bootstrap::bootstrap / bootstrap_without_key_switch
 * (src/bootstrap/vanilla.rs:40-63) and LutBootstrap::bootstrap_lut
 * (src/bootstrap/lut.rs:79-99), mapped over a batch.
 *   in        [count][n+1]
 *   testvec   NULL = the cloud key's test vector; else a LookupTable.poly
 *             ([2][N], shared by the batch) or [count][2][N] when per_ct != 0
 *   keyswitch 1: sample_extract_index(.,0) + identity_key_switching
 *             0: sample_extract_index_2(.,0)  (vanilla.rs:54-63)
 *   out       [count][n+1] */
int tfhe_hip_batch_bootstrap(tfhe_hip_ctx *ctx, const uint32_t *in, const uint32_t *testvec,
                             int per_ct, int keyswitch, uint32_t *out, size_t count);
int tfhe_hip_batch_bootstrap_dev(tfhe_hip_ctx *ctx, const uint32_t *in, const uint32_t *testvec,
                                 int per_ct, int keyswitch, uint32_t *out, size_t count,
                                 void *stream);

/* Replaces: the TLWE arithmetic between bootstraps -- Add, Sub, Neg, AddMul, SubMul for &TLWELv0
 * (src/tlwe.rs:129-214) -- mapped over a batch:  out = ca * a + cb * b  on all n+1 words (wrapping u32),
 * then out[n] += cconst.  Add = (1, 1, 0), Sub = (1, -1, 0), Neg = (-1, 0, 0) with b = NULL,
 * AddMul(k) = (1, k, 0), SubMul(k) = (1, -k, 0).  b may be NULL when cb == 0.  out may alias a or b. */
int tfhe_hip_batch_tlwe_lincomb(tfhe_hip_ctx *ctx, uint32_t ca, const uint32_t *a, uint32_t cb,
                                const uint32_t *b, uint32_t cconst, uint32_t *out, size_t count);
int tfhe_hip_batch_tlwe_lincomb_dev(tfhe_hip_ctx *ctx, uint32_t ca, const uint32_t *a, uint32_t cb,
                                    const uint32_t *b, uint32_t cconst, uint32_t *out, size_t count,
                                    void *stream);

/* Replaces: `bootstrap_lut(&(&x + &y), &lut, &cloud_key)` -- a TLWE linear combination fed straight
 * into a (programmable) bootstrap, the shape of every step of the reference's LUT arithmetic
 * (examples/lut_add_two_numbers.rs:124-158, examples/lut_arithmetic_demo.rs) and, with the default
 * test vector, of every gate (src/gates.rs:54-150).  The combination is computed in the prologue of
 * the blind-rotation kernel; arguments as in tfhe_hip_batch_tlwe_lincomb and tfhe_hip_batch_bootstrap. */
int tfhe_hip_batch_lincomb_bootstrap(tfhe_hip_ctx *ctx, uint32_t ca, const uint32_t *a, uint32_t cb,
                                     const uint32_t *b, uint32_t cconst, const uint32_t *testvec,
                                     int per_ct, int keyswitch, uint32_t *out, size_t count);
int tfhe_hip_batch_lincomb_bootstrap_dev(tfhe_hip_ctx *ctx, uint32_t ca, const uint32_t *a, uint32_t cb,
                                         const uint32_t *b, uint32_t cconst, const uint32_t *testvec,
                                         int per_ct, int keyswitch, uint32_t *out, size_t count,
                                         void *stream);

/* Replaces: trgsw::batch_blind_rotate[_with_railgun] (src/trgsw.rs:289-305),
 * trgsw::blind_rotate (:198-226) and blind_rotate_with_testvec (:242-274).
 * in [count][n+1]; testvec NULL or [2][N]; out_trlwe [count][2][N]. */
int tfhe_hip_batch_blind_rotate(tfhe_hip_ctx *ctx, const uint32_t *in, const uint32_t *testvec,
                                uint32_t *out_trlwe, size_t count);
int tfhe_hip_batch_blind_rotate_dev(tfhe_hip_ctx *ctx, const uint32_t *in,
                                    const uint32_t *testvec, uint32_t *out_trlwe, size_t count,
                                    void *stream);

/* Replaces: Gates::mux (src/gates.rs:157-183, naive == 0; the reference
 * formula reproduced bit-for-bit) and Gates::mux_naive
 * (src/gates.rs:189-199, naive != 0), mapped over a batch.  [count][n+1] each. */
int tfhe_hip_batch_mux(tfhe_hip_ctx *ctx, int naive, const uint32_t *a, const uint32_t *b,
                       const uint32_t *c, uint32_t *out, size_t count);
int tfhe_hip_batch_mux_dev(tfhe_hip_ctx *ctx, int naive, const uint32_t *a, const uint32_t *b,
                           const uint32_t *c, uint32_t *out, size_t count, void *stream);

/* ---- single stages (parity tests; same kernels' device code) ------------ */

/* Replaces: trgsw::external_product_with_fft (src/trgsw.rs:77-116) with
 * trgsw_fft = cloud_key.bootstrapping_key[bsk_index[c]].
 * trlwe_in/out [count][2][N]; bsk_index [count]. */
int tfhe_hip_batch_external_product(tfhe_hip_ctx *ctx, const uint32_t *trlwe_in,
                                    const int32_t *bsk_index, uint32_t *trlwe_out, size_t count);

/* Replaces: trlwe::sample_extract_index(., k) (src/trlwe.rs:106-120), 0 <= k < N (the bootstrap uses
 * k = 0): out[i] = a[k-i] for i <= k, MAX - a[N+k-i] for i > k, out[N] = b[k].
 * trlwe [count][2][N] -> out [count][N+1]. */
int tfhe_hip_batch_sample_extract(tfhe_hip_ctx *ctx, const uint32_t *trlwe, int k, uint32_t *out,
                                  size_t count);

/* Replaces: trgsw::identity_key_switching (src/trgsw.rs:332-360).
 * tlwe_lv1 [count][N+1] -> out [count][n+1]. */
int tfhe_hip_batch_identity_key_switch(tfhe_hip_ctx *ctx, const uint32_t *tlwe_lv1,
                                       uint32_t *out, size_t count);

/* ---- proxy re-encryption (the reference's feature `proxy-reenc`, src/proxy_reenc.rs; SURVEY 8(f) rank 4: "same
 * digit-lookup kernel shape as keyswitch") -------------------------------------------------------------------------
 * Replaces: proxy_reenc::reencrypt_tlwe_lv0 (src/proxy_reenc.rs:468-510) over a batch:
 *   out.b = in.b;  for i < n, j < t:  k = ((in.a[i] + 2^(32-(1+basebit*t))) >> (32-(j+1)*basebit)) & (base-1);
 *                                     k != 0: out -= key[base*t*i + base*j + k]          (all n+1 words, wrapping)
 * i.e. trgsw::identity_key_switching with a source of n coefficients, and it runs on the same kernels (the source is
 * padded to N coefficients whose digits are all zero).  A context -- or a key view of one (tfhe_hip_key_create), so that
 * cloud keys and re-encryption keys stay resident side by side -- holds EITHER a cloud key OR a re-encryption key:
 * loading one drops the other; the bootstrap entry points return TFHE_HIP_ENOKEY on a handle that holds a
 * re-encryption key, and these return it on a handle that does not.
 *   key [n][t][base][n+1] u32 = ProxyReencryptionKey::key_encryptions (src/proxy_reenc.rs:224-233; the k = 0 entries
 *   are never read, as in the reference :311-313), with base = 2^basebit and t of the context's parameter set (the
 *   reference's new_symmetric / new_asymmetric use params::trgsw_lv1::{BASEBIT, IKS_T}, :271-279, :362-370; for
 *   *_with_params keys create the context with that basebit / t).  The key is generated by the client (it needs the
 *   delegator's secret key): rs-tfhe_amd/proxy_reenc.py mirrors PublicKeyLv0 and ProxyReencryptionKey.
 *   Needs n <= N = 1024 (every set but SECURITY_UINT5 .. 8): TFHE_HIP_EINVAL otherwise.
 *   in [count][n+1] -> out [count][n+1]; the _dev form takes device pointers and a hipStream_t (NULL = the context's)
 *   and only enqueues. */
int tfhe_hip_load_reenc_key(tfhe_hip_ctx *ctx, const uint32_t *key);
int tfhe_hip_reenc_key_is_loaded(tfhe_hip_ctx *ctx);
int tfhe_hip_batch_reencrypt(tfhe_hip_ctx *ctx, const uint32_t *in, uint32_t *out, size_t count);
int tfhe_hip_batch_reencrypt_dev(tfhe_hip_ctx *ctx, const uint32_t *in, uint32_t *out, size_t count, void *stream);

/* Replaces: FFTProcessor::{ifft, fft, poly_mul, batch_ifft, batch_fft}
 * (src/fft/mod.rs:80-107; KlemsaProcessor src/fft/klemsa.rs:88-174) and the
 * SPQLIOS C ABI Spqlios_ifft_lv1 / _fft_lv1 / _poly_mul_1024
 * (spqlios-wrapper.cpp:19-40).  Spectra use the Klemsa layout and scaling.
 *   ifft:     src [count][N] u32 -> res [count][N] f64
 *   fft:      src [count][N] f64 -> res [count][N] u32
 *   poly_mul: a, b [count][N] u32 -> res [count][N] u32  (negacyclic) */
int tfhe_hip_batch_ifft(tfhe_hip_ctx *ctx, double *res, const uint32_t *src, size_t count);
int tfhe_hip_batch_fft(tfhe_hip_ctx *ctx, uint32_t *res, const double *src, size_t count);
int tfhe_hip_batch_poly_mul(tfhe_hip_ctx *ctx, uint32_t *res, const uint32_t *a,
                            const uint32_t *b, size_t count);

/* ---- concurrent callers --------------------------------------------------- */

/* Replaces: the concurrency a `Send + Sync` strategy gets from Rayon's workers (src/bootstrap/mod.rs:23-38,
 * src/parallel/rayon_impl.rs:40-47).  Host-pointer calls of at most `max_count` ciphertexts are merged with the calls
 * other threads make meanwhile (see "Thread safety" above); larger calls run alone, as before.  The default bound is
 * twice the device's CU count (512: what one launch of the latency kernels takes; a lone caller pays nothing up to
 * there, concurrent callers of 512-gate calls gain 23 %); 0 switches merging off, the largest bound is 4096.  The
 * environment variable TFHE_HIP_COMBINE (a number, 0 = off), read when a context is created, sets the same bound.
 * Accepts a context or a key view (the setting is the context's).
 * Bulk work yields to small calls: a merged launch needs whole CUs and cannot start while a launch of tens of thousands
 * of ciphertexts holds them all, so while small calls have been arriving (within the last 250 ms) the batch kernel of a
 * large call on the same context goes out in launches of 8,192 ciphertexts -- a small call then waits for a chunk
 * boundary (about 40 ms) instead of the whole batch (300 ms), at 1.7 % of the batch's throughput.  Same results; with no
 * small calls in flight a batch is one launch, as before. */
int tfhe_hip_set_combining(tfhe_hip_ctx *ctx, size_t max_count);

typedef struct tfhe_hip_combine_stats {
  uint64_t max_count;               /* the bound in force                                         */
  uint64_t launches;                /* merged launches issued                                     */
  uint64_t requests;                /* calls they carried                                         */
  uint64_t ciphertexts;             /* ciphertexts they carried                                   */
  uint64_t max_requests_per_launch; /* most calls taken by one leader at once                     */
  uint64_t lingers;                 /* leaders that waited for the previous launch's callers      */
  double linger_us;                 /* ... and for how long in all                                */
  double pack_us;                   /* leaders' time, summed: packing operands into the arena     */
  double gpu_us;                    /* ... copies in, kernels, copy out, the synchronise          */
  double unpack_us;                 /* ... handing every caller its rows                          */
} tfhe_hip_combine_stats;
/* Counters since the last call; resets them. */
int tfhe_hip_get_combine_stats(tfhe_hip_ctx *ctx, tfhe_hip_combine_stats *out);

/* ---- measurement -------------------------------------------------------- */

/* When enabled, every blind-rotate / key-switch launch is bracketed by HIP
 * events on the stream it is launched on. */
int tfhe_hip_set_profiling(tfhe_hip_ctx *ctx, int enabled);

typedef struct tfhe_hip_kernel_times {
  double blind_rotate_ms; /* sum of blind-rotate kernel durations       */
  double key_switch_ms;   /* sum of key-switch kernel durations         */
  uint64_t blind_rotate_launches;
  uint64_t key_switch_launches;
  uint64_t bootstraps;    /* ciphertexts that went through blind rotate */
} tfhe_hip_kernel_times;

/* Synchronises the recorded events, returns the sums since the last reset and
 * resets them. */
int tfhe_hip_get_kernel_times(tfhe_hip_ctx *ctx, tfhe_hip_kernel_times *out);

/* Shader clock actually sustained by the blind-rotation kernel: while profiling is enabled every
 * workgroup adds its s_memtime (shader cycles) and s_memrealtime (constant-rate counter) deltas to a
 * device counter; shader_mhz = cycles / ticks * rtc_mhz over all workgroups since the last call.
 * Synchronises, returns the sample and resets it.  (Needed to price the kernel against the FP64
 * issue roofline at the clock the chip really ran, not at the 2.4 GHz datasheet maximum.) */
typedef struct tfhe_hip_clock_sample {
  double shader_mhz;       /* 0 when nothing was sampled */
  double rtc_mhz;          /* rate of the constant counter (hipDeviceAttributeWallClockRate) */
  uint64_t shader_cycles;  /* summed over workgroups */
  uint64_t rtc_ticks;
} tfhe_hip_clock_sample;
int tfhe_hip_get_clock_sample(tfhe_hip_ctx *ctx, tfhe_hip_clock_sample *out);
/* The same sample for the matrix-core key switch (k_key_switch_mfma; the other key-switch kernels do not
 * sample): the clock that prices it against the int8 MFMA roofline. */
int tfhe_hip_get_key_switch_clock_sample(tfhe_hip_ctx *ctx, tfhe_hip_clock_sample *out);

/* Which kernels a batch of `count` ciphertexts runs on under this handle's dispatch rules (the cloud key must be
 * loaded: the base-4 matrix-core key switch exists only once its byte planes do), as one line of text:
 *   "blind_rotate=batch[0,1024)+single[1024,1100) key_switch=mfma(k=4)"
 * blind_rotate: up to two launches over contiguous parts of the batch, each `batch` (k_blind_rotate, one wave per
 * ciphertext), `single` (k_blind_rotate_wide2, one eight-wave workgroup per ciphertext) or `pair`
 * (k_blind_rotate_pair); key_switch: `mfma` / `sliced` / `b4` / `generic` / `split` with the number of partial
 * walks merged by integer atomics (k) and, for `sliced`, the accumulator sets per lane.  Every choice returns the
 * same result bits (tests/test_gpu_parity.py::test_dispatch_crossovers_bit_exact); the line exists so that tests,
 * profiles and bench.py can say WHICH kernels a number was measured on.  buf receives a NUL-terminated string.
 *
 * Environment variables read at tfhe_hip_ctx_create / tfhe_hip_pool_create -- the complete supported list:
 *   TFHE_HIP_BR_KERNEL = auto | batch | single | pair            (default auto) run that blind-rotation kernel at
 *                                                                 every batch size instead of picking per size
 *   TFHE_HIP_KS_KERNEL = auto | mfma | sliced | b4 | generic | split  (default auto) likewise for the key switch;
 *                        creation fails with TFHE_HIP_EINVAL when the parameter set cannot run the kernel named
 *   TFHE_HIP_POOL_RCCL = 0 | 1 | 2    pools: 0 never use RCCL (peer copies only), 1 (default) RCCL when the pool's
 *                                      devices are distinct, 2 also for a pool of one (plumbing test)
 *   TFHE_HIP_POOL_PINNED_STAGING = 0 | 1   pools: stage pageable operands through per-member pinned arenas
 *                                      (default: 1 for pools of several members)
 * Anything else (crossover counts, chunk counts, ring depths) is compiled in; overrides for those exist only in
 * experiment builds (-DTFHE_EXPERIMENT, profiles/exp/build_variants.sh) and are not part of this interface. */
int tfhe_hip_describe_dispatch(tfhe_hip_ctx *ctx, size_t count, char *buf, size_t buflen);

/* How this context's blind-rotation kernels round the external product (KlemsaProcessor::fft's `round() as i64 as u32`,
 * src/fft/klemsa.rs:145-146): "fast" when the parameter set bounds the pre-rounding magnitude below 2^51
 * (2l * N * Bg/2 * 2^31: the 128 / 110 / 80-bit sets) and one add does it, "general" otherwise (|x| < 2^63: the
 * multiple of 2^32 is peeled off first; the l = 1 and l = 2 message-space sets).  Both give the reference's bits
 * wherever the f64 product is exact.  Chosen at tfhe_hip_ctx_create from (l, bgbit); constant for the context. */
const char *tfhe_hip_rounding_mode(const tfhe_hip_ctx *ctx);

/* Block until everything this context enqueued -- on its own stream and on the caller's stream of the
 * last *_dev call -- has finished.  Returns TFHE_HIP_EINVAL (and clears the condition) if a
 * tfhe_hip_batch_gates_mixed_dev launch since the last call met a gate code outside tfhe_hip_gate. */
int tfhe_hip_synchronize(tfhe_hip_ctx *ctx);

/* ---- several GPUs behind one handle --------------------------------------------------------------
 * Replaces: the Rayon `par_map` the reference's batch functions run on
 * (src/parallel/rayon_impl.rs:40-47, called from src/gates.rs:357-383, src/trgsw.rs:289-305): an
 * order-preserving, embarrassingly parallel map over the ciphertexts of a slice under one shared read-only
 * &CloudKey.  A pool is that map over devices: one context per entry of `devices` (an entry may repeat: two
 * contexts on one GPU), the cloud key generated / uploaded once on the first device and replicated device to
 * device in the engine layouts, the batch split contiguously over the first k = min(ndev, ceil(count / 256))
 * members (shard r of k = tfhe_hip_pool_shard; batches under 256 ciphertexts per device are not cut thinner: a
 * device runs that many in the time of one), one host thread per shard, results written into the caller's output
 * slice in input order.  No collective and no
 * exchange between devices on the data path.  Pool calls are serialised per pool; a member context borrowed
 * with tfhe_hip_pool_ctx() (for the *_dev entry points) must not be used while a pool call runs. */
typedef struct tfhe_hip_pool tfhe_hip_pool;

int tfhe_hip_pool_create(const tfhe_hip_params *params, const int *devices, int ndev, tfhe_hip_pool **out);
void tfhe_hip_pool_destroy(tfhe_hip_pool *pool);
int tfhe_hip_pool_size(const tfhe_hip_pool *pool);
/* How the pool's last cloud key reached the members: "rccl" (one grouped ncclBroadcast per key buffer over xGMI,
 * in place in the engine layouts; librccl is opened at run time; used when the pool's devices are distinct; the
 * communicator is created once per pool, on first use, and lives until tfhe_hip_pool_destroy) or
 * "peer-copy" (serial hipMemcpyPeer from member 0: the fallback, and what pools with a repeated device use).
 * TFHE_HIP_POOL_RCCL=0 disables the RCCL path. */
const char *tfhe_hip_pool_key_transport(const tfhe_hip_pool *pool);
/* A key view of a pool: one tfhe_hip_key_create view per member (same devices, streams and scratch), accepted by
 * every tfhe_hip_pool_* entry point; tfhe_hip_pool_destroy(view) frees only its keys.  Destroy views first. */
int tfhe_hip_pool_key_create(tfhe_hip_pool *pool, tfhe_hip_pool **key_view);
/* A member context, BORROWED: valid until tfhe_hip_pool_destroy, never to be passed to tfhe_hip_ctx_destroy. */
tfhe_hip_ctx *tfhe_hip_pool_ctx(tfhe_hip_pool *pool, int member);
/* Text of the last failed pool call ("device D: ..."), or of the last failed create when pool == NULL. */
const char *tfhe_hip_pool_last_error(const tfhe_hip_pool *pool);
/* [lo, hi) of shard `shard` of `nshards` over `count` items: contiguous, sizes differ by at most one, earlier
 * shards take the remainder (the split every pool batch call uses). */
void tfhe_hip_pool_shard(size_t count, int shard, int nshards, size_t *lo, size_t *hi);
/* Members a batch of `count` is actually spread over (= the nshards of its split): min(size, ceil(count / 256)),
 * at least 1 -- one device runs up to 256 ciphertexts in the time of one, so small batches are not cut thinner. */
int tfhe_hip_pool_members_for(const tfhe_hip_pool *pool, size_t count);

/* Cloud key for every member: same arguments and meaning as the tfhe_hip_*_cloud_key calls above. */
int tfhe_hip_pool_load_cloud_key(tfhe_hip_pool *pool, const double *bsk, const uint32_t *ksk, uint32_t decomp_offset,
                                 const uint32_t *testvec);
int tfhe_hip_pool_gen_cloud_key_secure(tfhe_hip_pool *pool, const uint32_t *key_lv0, const uint32_t *key_lv1,
                                       double alpha_ksk, double alpha_bsk);
int tfhe_hip_pool_gen_cloud_key_with_key(tfhe_hip_pool *pool, const uint32_t *key_lv0, const uint32_t *key_lv1,
                                         double alpha_ksk, double alpha_bsk, const uint8_t rng_key[32]);
int tfhe_hip_pool_gen_cloud_key(tfhe_hip_pool *pool, const uint32_t *key_lv0, const uint32_t *key_lv1,
                                double alpha_ksk, double alpha_bsk, uint64_t seed); /* TEST / BENCHMARK ONLY */
int tfhe_hip_pool_export_cloud_key(tfhe_hip_pool *pool, int member, double *bsk, uint32_t *ksk,
                                   uint32_t *decomp_offset, uint32_t *testvec);

/* The batched hot path over all members, HOST pointers: same arguments and semantics as the single-context host
 * entry points of the same name (gates.rs:352-547; gates.rs:157-199; bootstrap/{vanilla,lut}.rs; trgsw.rs:289-305;
 * tlwe.rs:129-214).  One host thread per shard; results land in the caller's output slice in input order.
 * Small calls (gate / gates_mixed[_nks] / bootstrap / lincomb_bootstrap / mux / blind_rotate of at most the combining bound, see tfhe_hip_set_combining)
 * made by concurrent threads do not queue on the pool: each goes to the member (one per distinct device) with the least
 * work queued and shares launches there with the other threads' calls. */
int tfhe_hip_pool_batch_gate(tfhe_hip_pool *pool, int gate, const uint32_t *a, const uint32_t *b, uint32_t *out,
                             size_t count);
int tfhe_hip_pool_batch_gates_mixed(tfhe_hip_pool *pool, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                                    uint32_t *out, size_t count);
int tfhe_hip_pool_batch_gates_mixed_nks(tfhe_hip_pool *pool, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                                        uint32_t *out, size_t count);
int tfhe_hip_pool_batch_bootstrap(tfhe_hip_pool *pool, const uint32_t *in, const uint32_t *testvec, int per_ct,
                                  int keyswitch, uint32_t *out, size_t count);
int tfhe_hip_pool_batch_tlwe_lincomb(tfhe_hip_pool *pool, uint32_t ca, const uint32_t *a, uint32_t cb, const uint32_t *b,
                                     uint32_t cconst, uint32_t *out, size_t count);
int tfhe_hip_pool_batch_lincomb_bootstrap(tfhe_hip_pool *pool, uint32_t ca, const uint32_t *a, uint32_t cb,
                                          const uint32_t *b, uint32_t cconst, const uint32_t *testvec, int per_ct,
                                          int keyswitch, uint32_t *out, size_t count);
int tfhe_hip_pool_batch_mux(tfhe_hip_pool *pool, int naive, const uint32_t *a, const uint32_t *b, const uint32_t *c,
                            uint32_t *out, size_t count);
int tfhe_hip_pool_batch_blind_rotate(tfhe_hip_pool *pool, const uint32_t *in, const uint32_t *testvec,
                                     uint32_t *out_trlwe, size_t count);

/* The same calls for a batch that is RESIDENT ON ONE MEMBER'S GPU.
 * Replaces: the same Rayon par_map (src/parallel/rayon_impl.rs:40-47, src/gates.rs:357-383) for a caller whose
 * ciphertexts already live in device memory -- the levels of a circuit, the output of a previous pool call -- and
 * that has no torch.distributed to shard them with: every operand and the result are DEVICE pointers on member
 * `home_member`'s GPU, and `stream` is a stream of that GPU (NULL = the member context's own).  The call only
 * ENQUEUES, like the single-context *_dev calls:
 *   - the batch is cut as tfhe_hip_pool_shard cuts it over tfhe_hip_pool_members_for(count) members; shard r runs on
 *     member (home_member + r) mod size, so shard 0 -- and any batch too small to cut -- never leaves home;
 *   - every other shard travels to its member and its result back by ONE grouped ncclSend / ncclRecv pair per
 *     operand and peer over the pool's persistent RCCL communicator (all xGMI links concurrently; shared operands
 *     such as one test vector for the batch are sent whole to every peer), or by hipMemcpyPeerAsync behind events
 *     when the pool has no communicator (a repeated device, no librccl, TFHE_HIP_POOL_RCCL=0) --
 *     tfhe_hip_pool_data_transport() says which ran;
 *   - members compute on their own streams, home on `stream`; work enqueued on `stream` after the call sees the
 *     complete result (the gather is ordered into `stream`).  Results are in input order.
 * Errors detected while enqueueing are returned; tfhe_hip_pool_synchronize() drains the members' streams and reports
 * device-side conditions (see tfhe_hip_synchronize).  The members must not be used through tfhe_hip_pool_ctx() by
 * other threads meanwhile.  gates (mixed forms) is a device pointer too. */
int tfhe_hip_pool_batch_gate_dev(tfhe_hip_pool *pool, int home_member, int gate, const uint32_t *a, const uint32_t *b,
                                 uint32_t *out, size_t count, void *stream);
int tfhe_hip_pool_batch_gates_mixed_dev(tfhe_hip_pool *pool, int home_member, const uint8_t *gates, const uint32_t *a,
                                        const uint32_t *b, uint32_t *out, size_t count, void *stream);
int tfhe_hip_pool_batch_gates_mixed_nks_dev(tfhe_hip_pool *pool, int home_member, const uint8_t *gates,
                                            const uint32_t *a, const uint32_t *b, uint32_t *out, size_t count,
                                            void *stream);
int tfhe_hip_pool_batch_bootstrap_dev(tfhe_hip_pool *pool, int home_member, const uint32_t *in, const uint32_t *testvec,
                                      int per_ct, int keyswitch, uint32_t *out, size_t count, void *stream);
int tfhe_hip_pool_batch_tlwe_lincomb_dev(tfhe_hip_pool *pool, int home_member, uint32_t ca, const uint32_t *a,
                                         uint32_t cb, const uint32_t *b, uint32_t cconst, uint32_t *out, size_t count,
                                         void *stream);
int tfhe_hip_pool_batch_lincomb_bootstrap_dev(tfhe_hip_pool *pool, int home_member, uint32_t ca, const uint32_t *a,
                                              uint32_t cb, const uint32_t *b, uint32_t cconst, const uint32_t *testvec,
                                              int per_ct, int keyswitch, uint32_t *out, size_t count, void *stream);
int tfhe_hip_pool_batch_mux_dev(tfhe_hip_pool *pool, int home_member, int naive, const uint32_t *a, const uint32_t *b,
                                const uint32_t *c, uint32_t *out, size_t count, void *stream);
int tfhe_hip_pool_batch_blind_rotate_dev(tfhe_hip_pool *pool, int home_member, const uint32_t *in,
                                         const uint32_t *testvec, uint32_t *out_trlwe, size_t count, void *stream);

/* Drain every member's own stream (what the *_dev pool calls enqueued on the members); the caller's `stream` is the
 * caller's to synchronise.  Reports the device-side conditions tfhe_hip_synchronize reports. */
int tfhe_hip_pool_synchronize(tfhe_hip_pool *pool);
/* How the last *_dev pool call moved its shards: "rccl", "peer-copy", or "none" (nothing left the home GPU). */
const char *tfhe_hip_pool_data_transport(const tfhe_hip_pool *pool);

/* Measurement: tfhe_hip_set_profiling on every member, plus HIP-event pairs around every shard transfer of the *_dev
 * pool calls.  tfhe_hip_pool_get_transfer_times synchronises those events, returns the sums since the last call and
 * resets them: *_ms_sum over all transfers, *_ms_max the longest single one (transfers to different peers overlap).
 * Each moved shard is bracketed on the member's stream (the receiver of a scatter, the sender of a gather; a
 * gather's bracket opens after the member's compute); on the RCCL path the call's group is bracketed on the home
 * stream too, and a shard's time is the SHORTER of the two: a transfer starts when both ends have reached it, so the
 * end that arrived last brackets the transfer alone and the other one also brackets its wait for the peer.  (When both
 * brackets live on one device -- a self send / receive -- their events are comparable and the time is exact: from the
 * later arrival to the later completion.)
 * comm_create_ms / key_replication_ms are host wall-clock set-up costs and are NOT reset by reading: the creation of
 * the pool's persistent communicator (0 when it has none) and the last replication of a cloud key to the members. */
typedef struct tfhe_hip_pool_transfer_times {
  double scatter_ms_sum, scatter_ms_max;
  double gather_ms_sum, gather_ms_max;
  uint64_t scatter_bytes, gather_bytes;
  uint64_t calls; /* *_dev pool calls since the last read */
  double comm_create_ms, key_replication_ms;
  /* RCCL path: the home stream's bracket around each call's WHOLE group (every peer's transfer; for a gather also the
   * wait for the slowest member's compute), summed over calls -- an upper bound of any one transfer in the group.  Under
   * RCCL between distinct devices a shard's figure above is bounded by it, so *_ms_sum can approach (members - 1) x this:
   * read per-link cost from *_ms_max and the byte counts, not from the sums.  0 on the peer-copy path. */
  double scatter_group_ms_sum, gather_group_ms_sum;
} tfhe_hip_pool_transfer_times;
int tfhe_hip_pool_set_profiling(tfhe_hip_pool *pool, int enabled);
int tfhe_hip_pool_get_transfer_times(tfhe_hip_pool *pool, tfhe_hip_pool_transfer_times *out);

#ifdef __cplusplus
}
#endif
#endif /* TFHE_HIP_H */
