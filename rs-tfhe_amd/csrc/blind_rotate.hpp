// blind_rotate.hpp -- the persistent blind-rotation kernel and its stage kernels.
//
// Mapping: one 64-lane wavefront (= one workgroup) owns one ciphertext for all
// n CMUX steps.  The TRLWE accumulator never leaves the CU: it sits in a
// wave-private 8 KiB LDS array in natural coefficient order; lane l works on
// coefficients {l+64m, l+64m+512 : m<8}, the distribution the folded FFT
// consumes and produces, so there is no repacking between steps and X^k * acc
// is an indexed re-read.  LDS per wave: 9216-byte FFT tile + 8192-byte
// accumulator + the n rotation amounts (~18.8 KiB -> 8 waves per CU).
//
// Reference semantics reproduced here (paths relative to the rs-tfhe repo):
//   blind_rotate / blind_rotate_with_testvec   src/trgsw.rs:198-226, 242-274
//   poly_mul_with_x_k (Torus::MAX - x quirk)   src/trgsw.rs:307-330
//   cmux                                       src/trgsw.rs:174-196
//   decomposition                              src/trgsw.rs:144-171
//   external_product_with_fft / fma_in_fd_1024 src/trgsw.rs:77-142
//   KlemsaProcessor::ifft / fft                src/fft/klemsa.rs:88-150
//   gate linear prep                           src/gates.rs:54-150
//   sample_extract_index / _2                  src/trlwe.rs:106-136
#pragma once
#include "fft512.hpp"

namespace tfhe {

// Bootstrapping key in engine order: [n][2l][2][8][64] complex (double2),
// element (i, r, c, s, mu) = reference bin bin_of(mu, s) of
// bootstrapping_key[i].trlwe_fft[r].{a,b}, times 2^-10 (exact).

// X^k * p evaluated at coefficient j (k in [0, 2N]), reading p from LDS/global:
// idx = (j - k) mod 2N; idx < N ? p[idx] : MAX - p[idx-N]   (trgsw.rs:315-327)
template <typename P>
__device__ __forceinline__ uint32_t rot_read(const P *p, int j, int k) {
  int idx = (j - k) & (2 * kN - 1);
  uint32_t v = p[idx & (kN - 1)];
  return (idx & kN) ? ~v : v;  // Torus::MAX - v == ~v
}

using f64x2 = __attribute__((ext_vector_type(2))) double;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// (TFHE_ABL_NOKEY: timing-only experiment switch, see experiment.hpp -- 0 in every product build)
__device__ __forceinline__ f64x2 ldkey(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_off, uint32_t soff) {
  if (TFHE_ABL_NOKEY) {
    f64x2 r;
    r.x = (double)(lane_off + soff);
    r.y = (double)(lane_off ^ soff);
    return r;
  }
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)soff, 0);
  return __builtin_bit_cast(f64x2, v);
}

// ---- the external-product core shared by every kernel --------------------------
// One half of an external product: the L decomposition digits of ONE polynomial of
// the TRLWE (half_sel 0: a -> key rows 0..L-1, 1: b -> rows L..2L-1), each digit
// polynomial through a forward FFT and multiply-accumulated against its key row.
// In:  t_lo/hi[m] = (coefficient + decomposition offset), folded-FFT distribution
//      (lane l: coefficients l+64m and l+64m+512).
// Acc: fa / fb = the two accumulated spectra (un-normalised; the key carries 2^-10)
//      in the forward-FFT bin order, ready for fft_inverse.  INIT0: the first row
//      WRITES the accumulators (no zero fill, no add).
//      (decomposition trgsw.rs:144-171, batch_ifft + fma_in_fd_1024 trgsw.rs:99-106)
// How many of the 8 a-half / b-half key loads of a row are issued before the forward FFT.
#ifndef TFHE_PREFETCH_A
#define TFHE_PREFETCH_A 8
#endif
#ifndef TFHE_PREFETCH_B
#define TFHE_PREFETCH_B 4
#endif
#ifndef TFHE_L1_PA  // cap on TFHE_PREFETCH_A at L == 1 (a whole a-half in flight spilled there in round 2)
#define TFHE_L1_PA 7
#endif
// 1 = the inverse-pass-3 twiddles (20 VGPRs) are re-read from the cache-resident table before the inverse
// transforms of each CMUX step instead of living in registers across the forward phase (measured: 432 vs 469 ms
// on the round-2 kernel before the workgroup change, profiles/exp/logs/r2a_ab_fft_v2_and_ablations.log)
#ifndef TFHE_RELOAD_I3
#define TFHE_RELOAD_I3 1
#endif
// 1 = the two inverse transforms of a CMUX step are interleaved through the one tile (fft_inverse2)
#ifndef TFHE_INV_PAIR
#define TFHE_INV_PAIR 1
#endif

// Signed bit-field extract (v_bfe_i32).  Written as inline asm on purpose: with
// __builtin_amdgcn_sbfe and a run-time width, hipcc (ROCm 7.2 / clang 22) turns the following
// int->double conversion into v_cvt_f64_U32 (it assumes the result non-negative), which
// silently corrupts every negative digit.  The asm statement is opaque to that fold.
__device__ __forceinline__ int32_t sbfe(uint32_t src, int shift, int width) {
  int32_t d;
  asm("v_bfe_i32 %0, %1, %2, %3" : "=v"(d) : "v"(src), "s"(shift), "v"(width));
  return d;
}

// Waves per workgroup of the batch kernel (each wave still owns one ciphertext and its own LDS regions) and
// how tightly a workgroup's waves are held together: 0 = free-running, 1 = one barrier per CMUX step,
// 2 = per polynomial half, 3 = per digit row.  Waves that walk the key together fetch each 1-KiB key slice
// within a few hundred cycles of each other, so all but the first are served by the CU's vector L1 instead
// of the L2 -- the L2 -> L1 key stream (68.8 MB per ciphertext) is what holds the shader clock down.
#ifndef TFHE_WG_WAVES
#define TFHE_WG_WAVES 4
#endif
#ifndef TFHE_WG_SYNC
#define TFHE_WG_SYNC 1
#endif
template <int LEVEL>
__device__ __forceinline__ void wg_sync() {
  if (TFHE_WG_WAVES > 1 && TFHE_WG_SYNC >= LEVEL) __builtin_amdgcn_s_barrier();
}

// acc[j] += v in LDS.  1 (default): one ds_add_u32 without return -- the add happens in the LDS, nothing travels
// to the registers and back (it was ds_read_b32 + v_add_u32 + ds_write_b32); a wave's LDS operations execute in
// order, so the next step's reads see it.  0: the read-modify-write through registers.
#ifndef TFHE_ACC_LDS_ADD
#define TFHE_ACC_LDS_ADD 1
#endif
__device__ __forceinline__ void acc_add(uint32_t *p, uint32_t v) {
#if TFHE_ACC_LDS_ADD
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  *p += v;
#endif
}

// f (+)= x * v, complex, as four fused multiply-adds (fma_in_fd_1024, trgsw.rs:118-142; the 0.5 of the
// reference is in the key's 2^-10).  Written with explicit fma(): from `f += xr*v.x - xi*v.y` the
// compiler makes mul + fma + add.
template <bool INIT>
__device__ __forceinline__ void cmac(double &fr, double &fi, double xr, double xi, f64x2 v) {
  if (INIT) {
    fr = xr * v.x;
    fi = xr * v.y;
  } else {
    fr = fma(xr, v.x, fr);
    fi = fma(xr, v.y, fi);
  }
  fr = fma(-xi, v.y, fr);
  fi = fma(xi, v.x, fi);
}

// One decomposition row: digit extraction, forward transform, multiply-accumulate against key row r.
template <int L, bool INIT>
__device__ __forceinline__ void external_product_row(int r, int shift, const uint32_t (&w_lo)[8],
                                                     const uint32_t (&w_hi)[8], __amdgpu_buffer_rsrc_t bsk_rsrc,
                                                     uint32_t bsk_i_off, const Twiddles &tw, double2 *tile,
                                                     int lane, int bgbit, double (&fa_re)[8], double (&fa_im)[8],
                                                     double (&fb_re)[8], double (&fb_im)[8]) {
  const uint32_t lane_off = (uint32_t)lane * 16u;
  wg_sync<3>();
  // key row r: 2 x 8 coalesced 16-byte loads per lane off one buffer descriptor (lane
  // offset in a VGPR, row offset in an SGPR: no per-lane address arithmetic), issued
  // ahead of the FFT they are consumed after, so their latency hides under it.
  const uint32_t row_off = bsk_i_off + (uint32_t)r * (2u * kN2 * 16u);
  f64x2 va[8], vb[8];
  // with a single digit row per half the schedule gets tighter: a whole a-half in flight spills there
  constexpr int PA = (L == 1 && TFHE_PREFETCH_A > TFHE_L1_PA) ? TFHE_L1_PA : TFHE_PREFETCH_A;
  constexpr int PB = TFHE_PREFETCH_B;
#pragma unroll
  for (int s = 0; s < PA; ++s) va[s] = ldkey(bsk_rsrc, lane_off, row_off + (uint32_t)s * 1024u);
#pragma unroll
  for (int s = 0; s < PB; ++s) vb[s] = ldkey(bsk_rsrc, lane_off, row_off + (uint32_t)(kN2 * 16 + s * 1024));
  double re[8], im[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    re[m] = (double)sbfe(w_lo[m], shift, bgbit);
    im[m] = (double)sbfe(w_hi[m], shift, bgbit);
  }
  fft_forward(re, im, tw, tile, lane);
  // the rest of the row is fetched behind the first MACs
#pragma unroll
  for (int s = PA; s < 8; ++s) va[s] = ldkey(bsk_rsrc, lane_off, row_off + (uint32_t)s * 1024u);
#pragma unroll
  for (int s = PB; s < 8; ++s) vb[s] = ldkey(bsk_rsrc, lane_off, row_off + (uint32_t)(kN2 * 16 + s * 1024));
#pragma unroll
  for (int s = 0; s < 8; ++s) cmac<INIT>(fa_re[s], fa_im[s], re[s], im[s], va[s]);
#pragma unroll
  for (int s = 0; s < 8; ++s) cmac<INIT>(fb_re[s], fb_im[s], re[s], im[s], vb[s]);
}

template <int L, bool INIT0>
__device__ __forceinline__ void external_product_half(int half_sel, const uint32_t (&t_lo)[8],
                                                      const uint32_t (&t_hi)[8],
                                                      __amdgpu_buffer_rsrc_t bsk_rsrc, uint32_t bsk_i_off,
                                                      const Twiddles &tw, double2 *tile, int lane, int bgbit,
                                                      uint32_t signmask, double (&fa_re)[8],
                                                      double (&fa_im)[8], double (&fb_re)[8],
                                                      double (&fb_im)[8]) {
  // digit_i = ((t >> shift_i) & (Bg-1)) - Bg/2  (trgsw.rs:162) = the sign-extended bgbit-wide
  // field of t ^ sum_i (Bg/2 << shift_i): flipping a field's top bit is subtracting Bg/2 mod Bg.
  // That sum is the decomposition offset itself (key.rs:78-89), so one XOR per coefficient
  // turns every digit into a single signed bit-field extract.
  uint32_t w_lo[8], w_hi[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    w_lo[m] = t_lo[m] ^ signmask;
    w_hi[m] = t_hi[m] ^ signmask;
  }
  if (TFHE_WG_SYNC == 2) wg_sync<2>();
  external_product_row<L, INIT0>(half_sel * L, 32 - bgbit, w_lo, w_hi, bsk_rsrc, bsk_i_off, tw, tile, lane, bgbit,
                                 fa_re, fa_im, fb_re, fb_im);
  // The remaining rows stay a LOOP: with the body duplicated (or, at L = 2, the one-trip loop flattened)
  // the scheduler overlaps rows and spills; the trip count is hidden from it for that reason.
  int rows = L;
  if (L > 1) asm volatile("" : "+s"(rows));
#pragma unroll 1
  for (int i = 1; i < rows; ++i)
    external_product_row<L, false>(half_sel * L + i, 32 - (i + 1) * bgbit, w_lo, w_hi, bsk_rsrc, bsk_i_off, tw, tile,
                                   lane, bgbit, fa_re, fa_im, fb_re, fb_im);
}

struct BlindRotateArgs {
  // inputs: prepared = ca*a + cb*b (wrapping), prepared[n] += cconst   (gates.rs:54-150)
  const uint32_t *in_a;  // [count][n+1]
  const uint32_t *in_b;  // [count][n+1] or nullptr when cb == 0
  uint32_t ca, cb, cconst;
  const uint8_t *gate_codes;  // optional [count]: per-ciphertext tfhe_hip_gate overriding ca/cb/cconst
  const uint32_t *testvec;  // [2][N] (per_ct_stride == 0) or [count][2][N]
  size_t per_ct_stride;     // in u32 elements: 0 or 2N
  const double2 *bsk;       // engine order
  const double2 *tw;        // twiddle table
  int n, bgbit;
  uint32_t offset;
  // outputs (any may be null)
  uint32_t *out_trlwe;  // [count][2][N]
  uint32_t *out_lv1;    // [count][N+1]  sample_extract_index(.,0)
  uint32_t *out_ext2;   // [count][n+1]  sample_extract_index_2(.,0)
  size_t count;         // ciphertexts of this launch (the last workgroup may be partly filled)
  // diagnostics (may be null)
  unsigned long long *clk;  // [2]: += shader cycles (s_memtime) and += constant-rate ticks (s_memrealtime) per workgroup
  uint32_t *err_flag;       // |= 1 when a gate code outside tfhe_hip_gate is seen (the ciphertext is then treated as COPY)
};

// src/gates.rs:54-150 as (ca, cb, const): prepared = ca*a + cb*b, prepared.b += const.
// Index = tfhe_hip_gate; constants are utils::f64_to_torus(+-1/8, +-1/4) (utils.rs:9-12).
__device__ constexpr uint32_t kGateCa[11] = {0xFFFFFFFFu, 1u, 1u, 1u, 1u, 0xFFFFFFFFu, 0xFFFFFFFFu, 1u, 0xFFFFFFFFu, 1u, 1u};
__device__ constexpr uint32_t kGateCb[11] = {0xFFFFFFFFu, 1u, 1u, 2u, 0xFFFFFFFEu, 0xFFFFFFFFu, 1u, 0xFFFFFFFFu, 1u, 0xFFFFFFFFu, 0u};
__device__ constexpr uint32_t kGateCc[11] = {0x20000000u, 0x20000000u, 0xE0000000u, 0x40000000u, 0xC0000000u, 0xE0000000u,
                                             0xE0000000u, 0xE0000000u, 0x20000000u, 0x20000000u, 0u};

// LDS per workgroup: per wave { FFT tile | accumulator (a then b, natural order) | rotation amounts }, then ONE
// pass-2 twiddle table for the whole workgroup (every wave stores the same 64 entries).  Two workgroups share a
// CU's 160 KiB up to n = 1183, i.e. for every parameter set (n <= 1160).
constexpr int kAccBytes = 2 * kN * 4;
constexpr int kBrWaves = TFHE_WG_WAVES;  // waves (= ciphertexts) per workgroup of k_blind_rotate
__host__ __device__ __forceinline__ size_t blind_rotate_wave_lds_bytes(int n) {
  return ((size_t)kTileBytes + kAccBytes + (size_t)n * 2 + 15) & ~(size_t)15;
}
__host__ __device__ __forceinline__ size_t blind_rotate_lds_bytes(int n) {
  return kBrWaves * blind_rotate_wave_lds_bytes(n) + kT2Bytes;
}
constexpr int kStageLdsBytes = kTileBytes + kT2Bytes;  // stage kernels: tile | T2 table

// The TRLWE accumulator lives in a wave-private LDS array for the whole n-step
// chain.  X^k * acc is then just an indexed re-read of that array (poly_mul_with_x_k
// never materialises), and the 32 VGPRs an in-register accumulator would pin across
// the eight FFTs are free, which is what lets two waves share a SIMD (<= 256 VGPRs).
template <int L, bool FAST>
__global__ __launch_bounds__(64 * kBrWaves, 2) void k_blind_rotate(BlindRotateArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_wg[];
  const int n = A.n;
  const int lane = threadIdx.x & 63;
  const int wave = kBrWaves > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  unsigned char *smem = smem_wg + (size_t)wave * blind_rotate_wave_lds_bytes(n);
  double2 *tile = reinterpret_cast<double2 *>(smem);
  uint32_t *acc = reinterpret_cast<uint32_t *>(smem + kTileBytes);
  uint16_t *s_abar = reinterpret_cast<uint16_t *>(smem + kTileBytes + kAccBytes);
  double2 *t2tab = reinterpret_cast<double2 *>(smem_wg + (size_t)kBrWaves * blind_rotate_wave_lds_bytes(n));
  // a partly filled last workgroup: the spare waves redo the last ciphertext (they take part in the
  // barriers) and store nothing
  size_t ct = (size_t)blockIdx.x * kBrWaves + wave;
  const bool live = ct < A.count;
  if (!live) ct = A.count - 1;
  const unsigned long long clk0 = A.clk ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rtc0 = A.clk ? __builtin_amdgcn_s_memrealtime() : 0ull;

  Twiddles tw;
  tw.load(A.tw, t2tab, lane);

  // ---- gate linear prep + rotation amounts ---------------------------------
  uint32_t gca = A.ca, gcb = A.cb, gcc = A.cconst;
  if (A.gate_codes) {  // mixed batch: this ciphertext's own gate (same table as the host's gate_prep)
    uint32_t code = A.gate_codes[ct];
    if (code > 10u) {  // not a tfhe_hip_gate: flag it (the host reports it at the next synchronising call)
      if (A.err_flag && lane == 0) atomicOr(A.err_flag, 1u);
      code = 10u;
    }
    gca = kGateCa[code];
    gcb = kGateCb[code];
    gcc = kGateCc[code];
  }
  const uint32_t *pa = A.in_a + ct * (size_t)(n + 1);
  const uint32_t *pb = (A.in_b && gcb) ? A.in_b + ct * (size_t)(n + 1) : nullptr;
  for (int i = lane; i < n; i += 64) {
    uint32_t p = gca * pa[i];
    if (pb) p += gcb * pb[i];
    // a_tilda = (p +wrap 2^20) >> 21   (trgsw.rs:210-211)
    s_abar[i] = (uint16_t)((uint32_t)(p + (1u << 20)) >> 21);
  }
  uint32_t pbody = gca * pa[n];
  if (pb) pbody += gcb * pb[n];
  pbody += gcc;
  // b_tilda = 2N - ((b as usize + 2^20) >> 21), no 32-bit wrap (trgsw.rs:202-203)
  const int b_tilda = 2 * kN - (int)(((uint64_t)pbody + (1ull << 20)) >> 21);

  // ---- acc = X^b_tilda * testvec -------------------------------------------
  const uint32_t *tv = A.testvec + ct * A.per_ct_stride;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    int j = lane + 64 * m;
    acc[j] = rot_read(tv, j, b_tilda);
    acc[kN + j] = rot_read(tv + kN, j, b_tilda);
  }
  __syncthreads();

  // ---- n sequential CMUXes: acc += BSK[i] (x) (X^a_tilda * acc - acc) ------
  constexpr uint32_t per_i_bytes = 2u * L * 2u * kN2 * 16u;  // one TRGSW in engine order
  const __amdgpu_buffer_rsrc_t bsk_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)A.bsk, 0, (int)((uint32_t)n * per_i_bytes), 0x00020000);
  const uint32_t offset = A.offset;
  uint32_t signmask = 0;  // sum_i (Bg/2) << (32 - (i+1)*bgbit): the top bit of every digit field
#pragma unroll
  for (int i = 0; i < L; ++i) signmask |= 1u << (32 - i * A.bgbit - 1);
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
    wg_sync<1>();  // (a barrier only every 2nd / 4th / 16th step: 722 / 713 / 698 M cycles and 331.9 / 332.1 / 333.4 ms vs 331.7)
    const int k = s_abar[i];
    double fa_re[8], fa_im[8], fb_re[8], fb_im[8];  // written by the first row of the a half
    // cmux: tmp = in2 - in1 = X^k*acc - acc (trgsw.rs:183-186), + decomposition offset; the a
    // half is consumed before the b half is formed, so only 16 of these are ever live
    {
      uint32_t t_lo[8], t_hi[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int j = lane + 64 * m;
        t_lo[m] = rot_read(acc, j, k) - acc[j] + offset;
        t_hi[m] = rot_read(acc, j + kN2, k) - acc[j + kN2] + offset;
      }
      external_product_half<L, true>(0, t_lo, t_hi, bsk_rsrc, (uint32_t)i * per_i_bytes, tw, tile, lane, A.bgbit,
                                     signmask, fa_re, fa_im, fb_re, fb_im);
    }
    {
      const uint32_t *p = acc + kN;
      uint32_t t_lo[8], t_hi[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int j = lane + 64 * m;
        t_lo[m] = rot_read(p, j, k) - p[j] + offset;
        t_hi[m] = rot_read(p, j + kN2, k) - p[j + kN2] + offset;
      }
      external_product_half<L, false>(1, t_lo, t_hi, bsk_rsrc, (uint32_t)i * per_i_bytes, tw, tile, lane, A.bgbit,
                                      signmask, fa_re, fa_im, fb_re, fb_im);
    }
#if TFHE_RELOAD_I3
    {
      int z = 0;
      asm volatile("" : "+v"(z));  // opaque 0: keeps the reload inside the loop (see Twiddles::reload_i3)
      tw.reload_i3(A.tw, lane, z);
    }
#endif
#if TFHE_INV_PAIR
    fft_inverse2(fa_re, fa_im, fb_re, fb_im, tw, tile, lane);
#pragma unroll
    for (int m = 0; m < 8; ++m) {  // res = ext + in1 (trgsw.rs:189-193)
      const int j = lane + 64 * m;
      acc_add(&acc[j], round_product<FAST>(fa_re[m]));
      acc_add(&acc[j + kN2], round_product<FAST>(fa_im[m]));
      acc_add(&acc[kN + j], round_product<FAST>(fb_re[m]));
      acc_add(&acc[kN + j + kN2], round_product<FAST>(fb_im[m]));
    }
#else
    fft_inverse(fa_re, fa_im, tw, tile, lane);
#pragma unroll
    for (int m = 0; m < 8; ++m) {  // res = ext + in1 (trgsw.rs:189-193)
      const int j = lane + 64 * m;
      acc[j] += round_product<FAST>(fa_re[m]);
      acc[j + kN2] += round_product<FAST>(fa_im[m]);
    }
    fft_inverse(fb_re, fb_im, tw, tile, lane);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int j = lane + 64 * m;
      acc[kN + j] += round_product<FAST>(fb_re[m]);
      acc[kN + j + kN2] += round_product<FAST>(fb_im[m]);
    }
#endif
    wave_lds_sync();  // the next step re-reads acc at rotated (other lanes') positions
  }

  // ---- epilogue (acc is final; any lane may read any coefficient) -------------
  if (!live) return;
  if (A.out_trlwe) {
    uint32_t *o = A.out_trlwe + ct * (size_t)(2 * kN);
#pragma unroll
    for (int m = 0; m < 32; ++m) o[lane + 64 * m] = acc[lane + 64 * m];
  }
  if (A.out_lv1) {
    // p[0]=a[0]; p[i]=MAX-a[N-i]; p[N]=b[0]   (trlwe.rs:106-120 with k=0)
    uint32_t *o = A.out_lv1 + ct * (size_t)(kN + 1);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int i = lane + 64 * m;
      o[i] = i == 0 ? acc[0] : ~acc[kN - i];
    }
    if (lane == 0) o[kN] = acc[kN];
  }
  if (A.out_ext2) {
    // same formula with N := n   (trlwe.rs:122-136 with k=0)
    uint32_t *o = A.out_ext2 + ct * (size_t)(n + 1);
    for (int i = lane; i < n; i += 64) o[i] = i == 0 ? acc[0] : ~acc[n - i];
    if (lane == 0) o[n] = acc[kN];
  }
  if (A.clk && lane == 0) {
    atomicAdd(&A.clk[0], __builtin_amdgcn_s_memtime() - clk0);
    atomicAdd(&A.clk[1], __builtin_amdgcn_s_memrealtime() - rtc0);
  }
}

// ---- TLWE arithmetic between bootstraps (tlwe.rs:129-214): out = ca*a + cb*b, out[n] += cconst -------
// Pure streaming integer work: one word per thread, coalesced, HBM-bound.
__global__ void k_tlwe_lincomb(uint32_t ca, const uint32_t *__restrict__ a, uint32_t cb,
                               const uint32_t *__restrict__ b, uint32_t cconst, uint32_t *__restrict__ out,
                               uint32_t width, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  uint32_t v = ca * a[idx];
  if (b) v += cb * b[idx];
  if (idx % width == width - 1) v += cconst;
  out[idx] = v;
}

// ---- stage kernels (parity tests; same device code) --------------------------

// external_product_with_fft (trgsw.rs:77-116): out = BSK[idx] (x) in
template <int L, bool FAST>
__global__ __launch_bounds__(64) void k_external_product(const uint32_t *in, const int32_t *bsk_index,
                                                          const double2 *bsk, uint32_t bsk_bytes,
                                                          const double2 *twt, int bgbit, uint32_t offset,
                                                          uint32_t *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *tile = reinterpret_cast<double2 *>(smem);
  const int lane = threadIdx.x;
  const size_t ct = blockIdx.x;
  Twiddles tw;
  tw.load(twt, reinterpret_cast<double2 *>(smem + kTileBytes), lane);
  constexpr uint32_t per_i_bytes = 2u * L * 2u * kN2 * 16u;
  const __amdgpu_buffer_rsrc_t bsk_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)bsk, 0, (int)bsk_bytes, 0x00020000);
  const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane(bsk_index[ct]);
  uint32_t signmask = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) signmask |= 1u << (32 - i * bgbit - 1);
  double fa_re[8], fa_im[8], fb_re[8], fb_im[8];
  {
    const uint32_t *p = in + ct * (size_t)(2 * kN);
    uint32_t t_lo[8], t_hi[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      t_lo[m] = p[lane + 64 * m] + offset;
      t_hi[m] = p[lane + 64 * m + kN2] + offset;
    }
    external_product_half<L, true>(0, t_lo, t_hi, bsk_rsrc, idx * per_i_bytes, tw, tile, lane, bgbit, signmask,
                                   fa_re, fa_im, fb_re, fb_im);
  }
  {
    const uint32_t *p = in + ct * (size_t)(2 * kN) + kN;
    uint32_t t_lo[8], t_hi[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      t_lo[m] = p[lane + 64 * m] + offset;
      t_hi[m] = p[lane + 64 * m + kN2] + offset;
    }
    external_product_half<L, false>(1, t_lo, t_hi, bsk_rsrc, idx * per_i_bytes, tw, tile, lane, bgbit, signmask,
                                    fa_re, fa_im, fb_re, fb_im);
  }
  uint32_t *o = out + ct * (size_t)(2 * kN);
  fft_inverse(fa_re, fa_im, tw, tile, lane);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    o[lane + 64 * m] = round_product<FAST>(fa_re[m]);
    o[lane + 64 * m + kN2] = round_product<FAST>(fa_im[m]);
  }
  fft_inverse(fb_re, fb_im, tw, tile, lane);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    o[kN + lane + 64 * m] = round_product<FAST>(fb_re[m]);
    o[kN + lane + 64 * m + kN2] = round_product<FAST>(fb_im[m]);
  }
}

// KlemsaProcessor::ifft (klemsa.rs:88-117): torus poly -> spectrum, reference layout + x2
__global__ __launch_bounds__(64) void k_ifft(const uint32_t *src, const double2 *twt, double *res) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *tile = reinterpret_cast<double2 *>(smem);
  const int lane = threadIdx.x;
  const size_t p = blockIdx.x;
  Twiddles tw;
  tw.load(twt, reinterpret_cast<double2 *>(smem + kTileBytes), lane);
  double re[8], im[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    re[m] = (double)(int32_t)src[p * kN + lane + 64 * m];         // `as i32 as f64` klemsa.rs:96
    im[m] = (double)(int32_t)src[p * kN + lane + 64 * m + kN2];   // :97
  }
  fft_forward(re, im, tw, tile, lane);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    int k = bin_of(lane, s);
    res[p * kN + k] = re[s] * 2.0;  // klemsa.rs:112-113
    res[p * kN + k + kN2] = im[s] * 2.0;
  }
}

// KlemsaProcessor::fft (klemsa.rs:119-150): spectrum (reference layout) -> torus poly
__global__ __launch_bounds__(64) void k_fft(const double *src, const double2 *twt, uint32_t *res) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *tile = reinterpret_cast<double2 *>(smem);
  const int lane = threadIdx.x;
  const size_t p = blockIdx.x;
  Twiddles tw;
  tw.load(twt, reinterpret_cast<double2 *>(smem + kTileBytes), lane);
  double re[8], im[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    int k = bin_of(lane, s);
    re[s] = src[p * kN + k] * 0x1p-10;  // 0.5 (:126) * 1/512 (:136), exact
    im[s] = src[p * kN + k + kN2] * 0x1p-10;
  }
  fft_inverse(re, im, tw, tile, lane);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    res[p * kN + lane + 64 * m] = round_half_away_to_torus(re[m]);  // f64::round, klemsa.rs:145-146
    res[p * kN + lane + 64 * m + kN2] = round_half_away_to_torus(im[m]);
  }
}

// KlemsaProcessor::poly_mul (klemsa.rs:152-174)
__global__ __launch_bounds__(64) void k_poly_mul(const uint32_t *a, const uint32_t *b,
                                                  const double2 *twt, uint32_t *res) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *tile = reinterpret_cast<double2 *>(smem);
  const int lane = threadIdx.x;
  const size_t p = blockIdx.x;
  Twiddles tw;
  tw.load(twt, reinterpret_cast<double2 *>(smem + kTileBytes), lane);
  double are[8], aim[8], bre[8], bim[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    are[m] = (double)(int32_t)a[p * kN + lane + 64 * m];
    aim[m] = (double)(int32_t)a[p * kN + lane + 64 * m + kN2];
    bre[m] = (double)(int32_t)b[p * kN + lane + 64 * m];
    bim[m] = (double)(int32_t)b[p * kN + lane + 64 * m + kN2];
  }
  fft_forward(are, aim, tw, tile, lane);
  fft_forward(bre, bim, tw, tile, lane);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    // (2A)(2B)*0.5 then 0.5/512 in the inverse = A*B/512
    double r = (are[s] * bre[s] - aim[s] * bim[s]) * 0x1p-9;
    double i = (are[s] * bim[s] + aim[s] * bre[s]) * 0x1p-9;
    are[s] = r;
    aim[s] = i;
  }
  fft_inverse(are, aim, tw, tile, lane);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    res[p * kN + lane + 64 * m] = round_half_away_to_torus(are[m]);
    res[p * kN + lane + 64 * m + kN2] = round_half_away_to_torus(aim[m]);
  }
}

// sample_extract_index(., k)  (trlwe.rs:106-120): [count][2][N] -> [count][N+1]
__global__ void k_sample_extract(const uint32_t *trlwe, int k, uint32_t *out, size_t count) {
  size_t ct = blockIdx.x;
  const uint32_t *a = trlwe + ct * (size_t)(2 * kN);
  uint32_t *o = out + ct * (size_t)(kN + 1);
  for (int i = threadIdx.x; i <= kN; i += blockDim.x) {
    if (i == kN)
      o[kN] = a[kN + k];  // b[k]
    else if (i <= k)
      o[i] = a[k - i];
    else
      o[i] = ~a[kN + k - i];  // Torus::MAX - a[N + k - i]
  }
}

// engine-order conversion of the bootstrapping key (upload time)
__global__ void k_bsk_convert(const double *ref, double2 *eng, size_t polys, double scale /* key_scale(fast) */) {
  // one block per polynomial spectrum (i, r, c); 512 threads
  size_t p = blockIdx.x;
  int t = threadIdx.x;  // engine position s*64 + mu
  int s = t >> 6, mu = t & 63;
  int k = bin_of(mu, s);
  eng[p * kN2 + t] = make_double2(ref[p * kN + k] * scale, ref[p * kN + k + kN2] * scale);
}

}  // namespace tfhe
