// blind_rotate_l1.hpp -- EXPERIMENT (not in the product library): the batch blind rotation for the l = 1 parameter sets
// (SECURITY_UINT2 .. UINT8) with THREE waves per SIMD -- VERDICT r3 task 4.  Built only into variants made with
//   profiles/exp/build_variants.sh <name> "-DTFHE_EXP_L1"
// Result (profiles/exp/logs/r4_ab_l1_three_waves.log): 166 VGPRs, no scratch, occupancy 3, the same bits as
// k_blind_rotate<1, .> -- and 198.9 ms against 195.7 for 65,536 SECURITY_UINT4 bootstraps: 2 % fewer shader cycles
// (446.6 M vs 456.1 M) at a clock 3.6 % lower (2,246 vs 2,330 MHz), because the l = 1 kernel, too, sits at the board's
// 1,400 W cap (1,341-1,350 W) and the half-tile transposes and table reloads cost energy.  Not adopted.
//
// k_blind_rotate<1, .> (blind_rotate.hpp) runs two waves per SIMD like its l = 3 sibling, but at l = 1 it is not held
// by the board's power: it sustains 2.24-2.35 GHz and its waves spend 22 % of their cycles stalled on LDS issue and
// 37 % not issuing at all (profiles/r3_uint4_pmc.csv) -- two transforms' worth of transposes per key row instead of six
// make it latency-bound.  A third wave per SIMD hides that, if it fits:
//   registers  <= 168 per lane.  The two spectral accumulators are 64, a transform's working set 32, the forward
//              twiddles 16.  The key row is therefore STREAMED: four of its sixteen 16-byte slices are in flight at a
//              time (16 registers) instead of the whole row (64); with three waves to cover for each other the deep
//              prefetch of the two-wave kernel is not needed.
//   LDS        twelve ciphertexts per CU: 13.3 KiB each.  The accumulator is 8 KiB, so the transpose tile is HALVED
//              (fft512.hpp, half-tile form: real parts, then imaginary parts through a 4.5 KiB tile -- same LDS
//              cycles, twice the instructions) and the n rotation amounts leave the LDS altogether: a_tilda of step
//              i + 1 is recomputed from the input ciphertext by two scalar loads issued at the top of step i
//              (wave-uniform: trgsw.rs:210-211 on gca * a[i] + gcb * b[i]).
// Arithmetic, operation order and rounding are those of k_blind_rotate<1, .>: the results are the same bits.
#pragma once
#include <type_traits>

#include "../../../rs-tfhe_amd/csrc/blind_rotate.hpp"

namespace tfhe {

// (Twiddles::reload_f3 as a free function: the forward-pass-3 constants re-read from the cache-resident table between
// pass 2 and pass 3, so that their 16 registers are live only across pass 3)
__device__ __forceinline__ void reload_f3(Twiddles &t, const double2 *__restrict__ tw, int lane, int opaque_zero) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double2 a = tw[q * 64 + lane + opaque_zero];
    t.f3[2 * q] = a.x;
    t.f3[2 * q + 1] = a.y;
  }
}

// ---- half-tile form: the same transposes, ONE 8-byte component at a time ---------------------------------------
// The tile is [8][72] doubles (4,608 B instead of 9,216): real parts go through, then imaginary parts.  Same index maps,
// same bank-conflict freedom (the maps are conflict-free for 8-byte elements too: b64 stores are serviced in 16
// contiguous lanes, b64 loads in 32), the same LDS cycles per transpose (16 b64 stores ~ 8 b128 stores, 16 b64 loads =
// 8 b128 loads) in twice the instructions.  What it buys is LDS CAPACITY: 12.5 KiB per ciphertext (tile + accumulator)
// instead of 17, i.e. twelve waves per CU instead of eight -- blind_rotate_l1.hpp.
constexpr int kHalfTileBytes = kTileCplx * 8;
__device__ __forceinline__ void tpA_write_h(const double (&v)[8], double *t, int lane) {
#pragma unroll
  for (int k = 0; k < 8; ++k) t[k * kPlane + lane] = v[k];
}
__device__ __forceinline__ void tpA_read_h(double (&v)[8], const double *t, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int s = 0; s < 8; ++s) v[s] = t[hi * kPlane + lo + 8 * s];
}
__device__ __forceinline__ void tpB_write_h(const double (&v)[8], double *t, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int k = 0; k < 8; ++k) t[hi * kPlane + k * 9 + lo] = v[k];
}
__device__ __forceinline__ void tpB_read_h(double (&v)[8], const double *t, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int s = 0; s < 8; ++s) v[s] = t[hi * kPlane + lo * 9 + s];
}
__device__ __forceinline__ void tpBi_write_h(const double (&v)[8], double *t, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int s = 0; s < 8; ++s) t[hi * kPlane + lo * 9 + s] = v[s];
}
__device__ __forceinline__ void tpBi_read_h(double (&v)[8], const double *t, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = t[hi * kPlane + k * 9 + lo];
}
__device__ __forceinline__ void tpAi_write_h(const double (&v)[8], double *t, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int s = 0; s < 8; ++s) t[hi * kPlane + lo + 8 * s] = v[s];
}
__device__ __forceinline__ void tpAi_read_h(double (&v)[8], const double *t, int lane) {
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = t[k * kPlane + lane];
}
// (a wave's LDS operations execute in program order: the stores of the imaginary parts cannot overtake the loads of
// the real parts issued before them, so the one tile serves both without a wait in between)
#define TFHE_TP_H(WR, RD, re, im, t, lane) \
  do {                                     \
    wave_lds_order();                      \
    WR(re, t, lane);                       \
    wave_lds_order();                      \
    RD(re, t, lane);                       \
    wave_lds_order();                      \
    WR(im, t, lane);                       \
    wave_lds_order();                      \
    RD(im, t, lane);                       \
  } while (0)

// (twt != nullptr: the pass-3 constants are re-read from the table just before pass 3 -- see Twiddles::reload_f3)
__device__ __forceinline__ void fft_forward_h(double (&re)[8], double (&im)[8], Twiddles &tw, double *tile, int lane,
                                              const double2 *twt = nullptr) {
  fwd_pass1(re, im);
  TFHE_TP_H(tpA_write_h, tpA_read_h, re, im, tile, lane);
  fwd_pass2(re, im, tw, lane);
  TFHE_TP_H(tpB_write_h, tpB_read_h, re, im, tile, lane);
  if (twt) {
    int z = 0;
#ifndef TFHE_FFT_HOST_EMU
    asm volatile("" : "+v"(z));  // opaque 0: keeps the reload where it is written
#endif
    reload_f3(tw, twt, lane, z);
  }
  fwd_pass3(re, im, tw);
}
// two inverse transforms interleaved through the one half tile (as fft_inverse2)
__device__ __forceinline__ void fft_inverse2_h(double (&xr)[8], double (&xi)[8], double (&yr)[8], double (&yi)[8],
                                               const Twiddles &tw, double *tile, int lane) {
  dft8<true>(xr, xi);
  TFHE_TP_H(tpBi_write_h, tpBi_read_h, xr, xi, tile, lane);
  dft8<true>(yr, yi);  // covers x's round trip
  TFHE_TP_H(tpBi_write_h, tpBi_read_h, yr, yi, tile, lane);
  inv_pass2(xr, xi, tw, lane);  // covers y's
  TFHE_TP_H(tpAi_write_h, tpAi_read_h, xr, xi, tile, lane);
  inv_pass2(yr, yi, tw, lane);
  TFHE_TP_H(tpAi_write_h, tpAi_read_h, yr, yi, tile, lane);
  inv_pass3(xr, xi, tw);
  inv_pass3(yr, yi, tw);
}



constexpr int kL1Waves = 4;  // waves (= ciphertexts) per workgroup; three workgroups per CU
__host__ __device__ __forceinline__ size_t blind_rotate_l1_wave_lds_bytes() { return (size_t)kHalfTileBytes + kAccBytes; }
__host__ __device__ __forceinline__ size_t blind_rotate_l1_lds_bytes() { return kL1Waves * blind_rotate_l1_wave_lds_bytes() + kT2Bytes; }

__device__ __forceinline__ uint32_t sload(const uint32_t *base, uint32_t byte_off) {  // s_load_dword, not yet waited for
  uint32_t v;
  asm volatile("s_load_dword %0, %1, %2" : "=s"(v) : "s"(base), "s"(byte_off) : "memory");
  return v;
}

template <bool FAST>
__global__ __launch_bounds__(64 * kL1Waves, 3) void k_blind_rotate_l1(BlindRotateArgs A) {
  constexpr int L = 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_wg[];
  const int n = A.n;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *smem = smem_wg + (size_t)wave * blind_rotate_l1_wave_lds_bytes();
  double *tile = reinterpret_cast<double *>(smem);
  uint32_t *acc = reinterpret_cast<uint32_t *>(smem + kHalfTileBytes);
  double2 *t2tab = reinterpret_cast<double2 *>(smem_wg + (size_t)kL1Waves * blind_rotate_l1_wave_lds_bytes());
  // a partly filled last workgroup: the spare waves redo the last ciphertext (they take part in the barriers) and store nothing
  size_t ct = (size_t)blockIdx.x * kL1Waves + wave;
  const bool live = ct < A.count;
  if (!live) ct = A.count - 1;
  const unsigned long long clk0 = A.clk ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rtc0 = A.clk ? __builtin_amdgcn_s_memrealtime() : 0ull;

  Twiddles tw;
  tw.load(A.tw, t2tab, lane);

  // ---- gate linear prep (gates.rs:54-150), wave-uniform ------------------------------------
  uint32_t gca = A.ca, gcb = A.cb, gcc = A.cconst;
  if (A.gate_codes) {
    uint32_t code = A.gate_codes[ct];
    if (code > 10u) {
      if (A.err_flag && lane == 0) atomicOr(A.err_flag, 1u);
      code = 10u;
    }
    gca = kGateCa[code];
    gcb = kGateCb[code];
    gcc = kGateCc[code];
  }
  gca = (uint32_t)__builtin_amdgcn_readfirstlane((int)gca);
  gcb = (uint32_t)__builtin_amdgcn_readfirstlane((int)gcb);
  const uint32_t *pa = A.in_a + ct * (size_t)(n + 1);
  const bool two = A.in_b && gcb;
  const uint32_t *pb = two ? A.in_b + ct * (size_t)(n + 1) : pa;  // (one operand: b is loaded too and multiplied by 0)
  if (!two) gcb = 0u;
  uint32_t pbody = gca * pa[n];
  if (two) pbody += gcb * pb[n];
  pbody += gcc;
  const int b_tilda = 2 * kN - (int)(((uint64_t)pbody + (1ull << 20)) >> 21);  // trgsw.rs:202-203

  // ---- acc = X^b_tilda * testvec ---------------------------------------------------------------
  const uint32_t *tv = A.testvec + ct * A.per_ct_stride;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int j = lane + 64 * m;
    acc[j] = rot_read(tv, j, b_tilda);
    acc[kN + j] = rot_read(tv + kN, j, b_tilda);
  }
  __syncthreads();

  constexpr uint32_t per_i_bytes = 2u * L * 2u * kN2 * 16u;  // one TRGSW in engine order
  const __amdgpu_buffer_rsrc_t bsk_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)A.bsk, 0, (int)((uint32_t)n * per_i_bytes), 0x00020000);
  const uint32_t offset = A.offset;
  const uint32_t signmask = 1u << 31;  // L = 1: the one digit field's top bit
  const int bgbit = A.bgbit, shift = 32 - bgbit;
  const uint32_t lane_off = (uint32_t)lane * 16u;

  // one half of the external product (trgsw.rs:77-116): digits of t, forward transform, multiply-accumulate against
  // key row `row`, the key streamed four slices at a time
  auto half = [&](auto init_c, const uint32_t *p, int k, uint32_t row_off, double (&fa_re)[8], double (&fa_im)[8],
                  double (&fb_re)[8], double (&fb_im)[8]) {
    constexpr bool INIT = decltype(init_c)::value;
    f64x2 ka[2][2], kb[2][2];  // [buffer][slice pair]
    auto fetch = [&](int s0, int buf) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        ka[buf][q] = ldkey(bsk_rsrc, lane_off, row_off + (uint32_t)(s0 + q) * 1024u);
        kb[buf][q] = ldkey(bsk_rsrc, lane_off, row_off + (uint32_t)(kN2 * 16 + (s0 + q) * 1024));
      }
    };
    fetch(0, 0);  // under the transform
    double re[8], im[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int j = lane + 64 * m;
      const uint32_t w_lo = (rot_read(p, j, k) - p[j] + offset) ^ signmask;  // cmux: X^k acc - acc (trgsw.rs:183-186)
      const uint32_t w_hi = (rot_read(p, j + kN2, k) - p[j + kN2] + offset) ^ signmask;
      re[m] = (double)sbfe(w_lo, shift, bgbit);
      im[m] = (double)sbfe(w_hi, shift, bgbit);
    }
    fft_forward_h(re, im, tw, tile, lane, A.tw);
#pragma unroll
    for (int s0 = 0; s0 < 8; s0 += 2) {
      __builtin_amdgcn_sched_barrier(0);
      if (s0 + 2 < 8) fetch(s0 + 2, ((s0 >> 1) + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        cmac<INIT>(fa_re[s0 + q], fa_im[s0 + q], re[s0 + q], im[s0 + q], ka[(s0 >> 1) & 1][q]);
        cmac<INIT>(fb_re[s0 + q], fb_im[s0 + q], re[s0 + q], im[s0 + q], kb[(s0 >> 1) & 1][q]);
      }
    }
  };

  // a_tilda of step 0, and the raw words of step 1 in flight
  uint32_t k_now;
  {
    const uint32_t a0 = pa[0], b0 = pb[0];
    k_now = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((gca * a0 + gcb * b0) + (1u << 20)) >> 21));  // trgsw.rs:210-211
  }
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
    __builtin_amdgcn_s_barrier();  // the workgroup's waves walk the key together (L1 hits), as in k_blind_rotate
    const int k = (int)k_now;
    // next step's words: two scalar loads that return under this step (clamped on the last step)
    const uint32_t nxt = (uint32_t)(i + 1 < n ? i + 1 : i) * 4u;
    const uint32_t na = sload(pa, nxt), nb = sload(pb, nxt);
    double fa_re[8], fa_im[8], fb_re[8], fb_im[8];
    half(std::true_type{}, acc, k, (uint32_t)i * per_i_bytes, fa_re, fa_im, fb_re, fb_im);
    half(std::false_type{}, acc + kN, k, (uint32_t)i * per_i_bytes + 2u * kN2 * 16u, fa_re, fa_im, fb_re, fb_im);
    {
      int z = 0;
      asm volatile("" : "+v"(z));  // opaque 0: keeps the reload inside the loop (see Twiddles::reload_i3)
      tw.reload_i3(A.tw, lane, z);
    }
    fft_inverse2_h(fa_re, fa_im, fb_re, fb_im, tw, tile, lane);
#pragma unroll
    for (int m = 0; m < 8; ++m) {  // res = ext + in1 (trgsw.rs:189-193)
      const int j = lane + 64 * m;
      acc_add(&acc[j], round_product<FAST>(fa_re[m]));
      acc_add(&acc[j + kN2], round_product<FAST>(fa_im[m]));
      acc_add(&acc[kN + j], round_product<FAST>(fb_re[m]));
      acc_add(&acc[kN + j + kN2], round_product<FAST>(fb_im[m]));
    }
    // lgkmcnt(0): the accumulator is final for this step (the next one re-reads it at other lanes' positions) AND the two
    // scalar loads have returned; their registers pass through the statement so that nothing reads them above it
    uint32_t ra = na, rb = nb;
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ra), "+s"(rb)::"memory");
    k_now = ((gca * ra + gcb * rb) + (1u << 20)) >> 21;
  }

  // ---- epilogue (as k_blind_rotate) ------------------------------------------------------------
  if (!live) return;
  if (A.out_trlwe) {
    uint32_t *o = A.out_trlwe + ct * (size_t)(2 * kN);
#pragma unroll
    for (int m = 0; m < 32; ++m) o[lane + 64 * m] = acc[lane + 64 * m];
  }
  if (A.out_lv1) {  // trlwe.rs:106-120 with k = 0
    uint32_t *o = A.out_lv1 + ct * (size_t)(kN + 1);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int i = lane + 64 * m;
      o[i] = i == 0 ? acc[0] : ~acc[kN - i];
    }
    if (lane == 0) o[kN] = acc[kN];
  }
  if (A.out_ext2) {  // trlwe.rs:122-136 with k = 0
    uint32_t *o = A.out_ext2 + ct * (size_t)(n + 1);
    for (int i = lane; i < n; i += 64) o[i] = i == 0 ? acc[0] : ~acc[n - i];
    if (lane == 0) o[n] = acc[kN];
  }
  if (A.clk && lane == 0) {
    atomicAdd(&A.clk[0], __builtin_amdgcn_s_memtime() - clk0);
    atomicAdd(&A.clk[1], __builtin_amdgcn_s_memrealtime() - rtc0);
  }
}

}  // namespace tfhe
