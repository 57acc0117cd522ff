// blind_rotate_wide1.hpp -- the round-1/2 latency kernel (one wave per decomposition row), SUPERSEDED by
// k_blind_rotate_wide2 (rs-tfhe_amd/csrc/blind_rotate_wide.hpp) and no longer part of the product library.
// Kept as the comparison point of profiles/exp/latency.py and latency_ablation.py: compiled only into experiment
// variants built with  profiles/exp/build_variants.sh <name> "-DTFHE_EXP_WIDE1"  (which passes -DTFHE_EXPERIMENT);
// such a library runs it for the SINGLE blind-rotation kernel when TFHE_HIP_BR_WIDE2=0 is set.  On the inexact
// parameter sets its 2L-term sums are ordered differently from the three shipped kernels, so its low bits differ
// there -- one reason it was removed from the product.
#pragma once
#include "../../../rs-tfhe_amd/csrc/blind_rotate.hpp"

namespace tfhe {

// ---- latency variant: one WORKGROUP of 2L waves per ciphertext ---------------------------------
// The batch kernel above is a throughput design: one wave walks all n steps alone (~17 ms per
// gate at SECURITY_128_BIT).  For batches smaller than the machine (count <= #CUs: the single-gate
// Bootstrap::bootstrap / Gates::nand calls of the reference API) this kernel spreads ONE ciphertext
// over 2L waves: wave r owns decomposition row r -- its digits, its forward FFT, its key row (a and
// b halves, prefetched one step ahead in registers) -- and publishes the two partial products in
// LDS; after one barrier waves 0 and 1 sum the 2L partials of the a / b spectrum, run the inverse
// FFT and update their half of the accumulator; a second barrier closes the step.  Same arithmetic
// per element as the batch kernel up to the order of the 2L-term sum, so results stay bit-exact
// wherever the products are exact.
__host__ __device__ __forceinline__ size_t blind_rotate_wide_lds_bytes(int n, int L) {
  return ((size_t)(2 * L + 2) * kTileBytes + (size_t)2 * L * kN2 * 16 + kAccBytes + kT2Bytes + (size_t)n * 2 + 15) &
         ~(size_t)15;
}

template <int L, bool FAST>
__global__ __launch_bounds__(128 * L, 1) void k_blind_rotate_wide(BlindRotateArgs A) {
  constexpr int W = 2 * L;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *tiles = reinterpret_cast<double2 *>(smem);  // [W + 2]: forward tiles (reused as fa partials) + 2 inverse tiles
  double2 *fbpart = reinterpret_cast<double2 *>(smem + (size_t)(W + 2) * kTileBytes);  // [W][512]
  uint32_t *acc = reinterpret_cast<uint32_t *>(smem + (size_t)(W + 2) * kTileBytes + (size_t)W * kN2 * 16);
  double2 *t2tab = reinterpret_cast<double2 *>(reinterpret_cast<unsigned char *>(acc) + kAccBytes);
  uint16_t *s_abar = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(t2tab) + kT2Bytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NT = 64 * W;
  const size_t ct = blockIdx.x;
  const int n = A.n;
  const unsigned long long clk0 = A.clk ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rtc0 = A.clk ? __builtin_amdgcn_s_memrealtime() : 0ull;

  Twiddles tw;
  tw.load(A.tw, t2tab, lane);  // every wave stores the same 64 entries; ends with a workgroup barrier

  uint32_t gca = A.ca, gcb = A.cb, gcc = A.cconst;
  if (A.gate_codes) {
    uint32_t code = A.gate_codes[ct];
    if (code > 10u) {
      if (A.err_flag && tid == 0) atomicOr(A.err_flag, 1u);
      code = 10u;
    }
    gca = kGateCa[code];
    gcb = kGateCb[code];
    gcc = kGateCc[code];
  }
  const uint32_t *pa = A.in_a + ct * (size_t)(n + 1);
  const uint32_t *pb = (A.in_b && gcb) ? A.in_b + ct * (size_t)(n + 1) : nullptr;
  for (int i = tid; i < n; i += NT) {
    uint32_t p = gca * pa[i];
    if (pb) p += gcb * pb[i];
    s_abar[i] = (uint16_t)((uint32_t)(p + (1u << 20)) >> 21);  // trgsw.rs:210-211
  }
  uint32_t pbody = gca * pa[n];
  if (pb) pbody += gcb * pb[n];
  pbody += gcc;
  const int b_tilda = 2 * kN - (int)(((uint64_t)pbody + (1ull << 20)) >> 21);  // trgsw.rs:202-203
  const uint32_t *tv = A.testvec + ct * A.per_ct_stride;
  for (int j = tid; j < kN; j += NT) {
    acc[j] = rot_read(tv, j, b_tilda);
    acc[kN + j] = rot_read(tv + kN, j, b_tilda);
  }
  __syncthreads();

  const int half_sel = wave / L, d = wave % L;
  const int bgbit = A.bgbit;
  const int shift = 32 - (d + 1) * bgbit;
  uint32_t signmask = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) signmask |= 1u << (32 - i * bgbit - 1);
  const uint32_t offset = A.offset;
  double2 *mytile = tiles + (size_t)wave * kTileCplx;
  double2 *fb_mine = fbpart + (size_t)wave * kN2;
  constexpr uint32_t per_i_bytes = 2u * L * 2u * kN2 * 16u;
  const __amdgpu_buffer_rsrc_t bsk_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)A.bsk, 0, (int)((uint32_t)n * per_i_bytes), 0x00020000);
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t my_row = (uint32_t)wave * (2u * kN2 * 16u);

  f64x2 va[8], vb[8];  // this wave's key row of the CURRENT step (loaded one step ahead)
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    va[s] = ldkey(bsk_rsrc, lane_off, my_row + (uint32_t)s * 1024u);
    vb[s] = ldkey(bsk_rsrc, lane_off, my_row + (uint32_t)(kN2 * 16 + s * 1024));
  }

#if defined(TFHE_EXPERIMENT) && defined(TFHE_LAT_STAMPS)  // per-phase cycle stamps of step n/2 (profiles/exp/latency.py --stamps)
#define LAT_STAMP(q) do { if (A.clk && i == n / 2 && lane == 0 && blockIdx.x == 0) A.clk[8 + wave * 16 + (q)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LAT_STAMP(q) do { } while (0)
#endif
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
    LAT_STAMP(0);
    const int k = s_abar[i];
    const uint32_t *p = acc + half_sel * kN;
    double re[8], im[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {  // tmp = X^k*acc - acc (+ offset), digit d of it
      const int j = lane + 64 * m;
      if (TFHE_ABL_LAT & 2) {  // timing-only: no rotated reads, no digit extraction
        re[m] = (double)(k + m);
        im[m] = (double)(lane - m);
        continue;
      }
      const uint32_t w_lo = (rot_read(p, j, k) - p[j] + offset) ^ signmask;
      const uint32_t w_hi = (rot_read(p, j + kN2, k) - p[j + kN2] + offset) ^ signmask;
      re[m] = (double)sbfe(w_lo, shift, bgbit);
      im[m] = (double)sbfe(w_hi, shift, bgbit);
    }
    LAT_STAMP(1);
    fft_forward(re, im, tw, mytile, lane);
    wave_lds_sync();  // the transform's last tile reads are done: the tile becomes the fa-partial slot
    LAT_STAMP(2);
    double2 keep_a[8], keep_b[8];  // (timing-only ablation bit 0: the products stay in registers, nothing is exchanged)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const double2 pa_ = make_double2(re[s] * va[s].x - im[s] * va[s].y, re[s] * va[s].y + im[s] * va[s].x);
      const double2 pb_ = make_double2(re[s] * vb[s].x - im[s] * vb[s].y, re[s] * vb[s].y + im[s] * vb[s].x);
      if (TFHE_ABL_LAT & 1) {
        keep_a[s] = pa_;
        keep_b[s] = pb_;
      } else {
        mytile[s * 64 + lane] = pa_;
        fb_mine[s * 64 + lane] = pb_;
      }
    }
    // next step's key row (clamped on the last step: a harmless re-read)
    const uint32_t nxt = (uint32_t)(i + 1 < n ? i + 1 : i) * per_i_bytes + my_row;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      va[s] = ldkey(bsk_rsrc, lane_off, nxt + (uint32_t)s * 1024u);
      vb[s] = ldkey(bsk_rsrc, lane_off, nxt + (uint32_t)(kN2 * 16 + s * 1024));
    }
    LAT_STAMP(3);
    __syncthreads();  // all 2L partial products of both spectra are in LDS
    LAT_STAMP(4);
    if (wave < 2) {   // wave 0: a spectrum (tiles), wave 1: b spectrum (fbpart)
      const double2 *src = wave == 0 ? tiles : fbpart;
      const size_t stride = wave == 0 ? (size_t)kTileCplx : (size_t)kN2;
      double f_re[8], f_im[8];
      if (TFHE_ABL_LAT & 1) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          f_re[s] = wave == 0 ? keep_a[s].x : keep_b[s].x;
          f_im[s] = wave == 0 ? keep_a[s].y : keep_b[s].y;
        }
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          double2 v = src[s * 64 + lane];
          f_re[s] = v.x;
          f_im[s] = v.y;
        }
#pragma unroll
        for (int w = 1; w < W; ++w)
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            double2 v = src[(size_t)w * stride + s * 64 + lane];
            f_re[s] += v.x;
            f_im[s] += v.y;
          }
      }
      LAT_STAMP(5);
      fft_inverse(f_re, f_im, tw, tiles + (size_t)(W + wave) * kTileCplx, lane);
      LAT_STAMP(6);
      uint32_t *q = acc + wave * kN;
#pragma unroll
      for (int m = 0; m < 8; ++m) {  // res = ext + in1 (trgsw.rs:189-193)
        const int j = lane + 64 * m;
        if (TFHE_ABL_LAT & 4) {  // timing-only: no rounding, no update (one store keeps the transform alive)
          if (m == 0 && f_re[0] + f_im[7] == 1.2345) q[j] = 1u;
          continue;
        }
        acc_add(&q[j], round_product<FAST>(f_re[m]));
        acc_add(&q[j + kN2], round_product<FAST>(f_im[m]));
      }
    }
    LAT_STAMP(7);
    __syncthreads();  // the accumulator is final for this step; partial slots are free again
    LAT_STAMP(8);
  }
#undef LAT_STAMP

  if (A.out_trlwe) {
    uint32_t *o = A.out_trlwe + ct * (size_t)(2 * kN);
    for (int j = tid; j < 2 * kN; j += NT) o[j] = acc[j];
  }
  if (A.out_lv1) {  // trlwe.rs:106-120 with k=0
    uint32_t *o = A.out_lv1 + ct * (size_t)(kN + 1);
    for (int i = tid; i < kN; i += NT) o[i] = i == 0 ? acc[0] : ~acc[kN - i];
    if (tid == 0) o[kN] = acc[kN];
  }
  if (A.out_ext2) {  // trlwe.rs:122-136 with k=0
    uint32_t *o = A.out_ext2 + ct * (size_t)(n + 1);
    for (int i = tid; i < n; i += NT) o[i] = i == 0 ? acc[0] : ~acc[n - i];
    if (tid == 0) o[n] = acc[kN];
  }
  if (A.clk && tid == 0) {
    atomicAdd(&A.clk[0], __builtin_amdgcn_s_memtime() - clk0);
    atomicAdd(&A.clk[1], __builtin_amdgcn_s_memrealtime() - rtc0);
  }
}

}  // namespace tfhe
