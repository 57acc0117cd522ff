// blind_rotate_wide.hpp -- the latency kernel, second form: one WORKGROUP OF EIGHT WAVES per ciphertext.
//
// k_blind_rotate_wide (blind_rotate.hpp) gives decomposition row r to wave r: digits, forward FFT, both products
// with its key row, 16 KiB of partial products through LDS, then waves 0 / 1 sum the 2l partials and run the inverse
// transforms.  Measured (profiles/exp/latency_ablation.py): 10,300 cycles per CMUX step, of which
// 2,220 are that exchange (the CU's two 39-B/cycle LDS store paths) and 1,950 the rotated reads + digit extraction that the l
// waves of a half all repeat.  This form removes both repetitions:
//   P0  the digit preparation is SHARED: four waves per half (its l forward waves + 4 - l of the waves without a row)
//       each compute w = ((X^k acc - acc) + offset) ^ signmask (trgsw.rs:183-186, 144-171) for a quarter of the
//       half's coefficients and publish it (4 KiB per half);
//   P1  wave r reads the 16 words it needs, extracts ITS digit, runs the forward FFT and publishes the SPECTRUM
//       (8 KiB, in its own transpose tile) -- not the products;
//   P2  the multiply-accumulate against the key is done per spectral SLOT: wave s (all eight waves) reads slot s of
//       the 2l spectra, holds slot s of every key row (prefetched a step ahead), accumulates fa and fb in the same
//       row order and with the same FMA sequence as the batch kernel -- so the sums are bit-identical to the batch
//       kernel's, for every parameter set, not only where the products are exact -- and publishes 2 KiB;
//   P3  waves 0 / 1 read the summed spectra, run the inverse transforms and update the accumulator.
// LDS stores per step fall from 224 KiB to 136 KiB, the 80-add partial sums disappear, the digit preparation is done
// once per half; the price is two more workgroup barriers per step.
#pragma once
#include "blind_rotate.hpp"

namespace tfhe {

constexpr int kWide2Waves = 8;

__host__ __device__ __forceinline__ size_t blind_rotate_wide2_lds_bytes(int n, int L) {
  // tiles of the 2L forward waves | summed spectra [2][512] | shared digit words [2][N] | accumulator | T2 | a_bar
  return ((size_t)(2 * L) * kTileBytes + (size_t)2 * kN2 * 16 + (size_t)2 * kN * 4 + kAccBytes + kT2Bytes + (size_t)n * 2 + 15) &
         ~(size_t)15;
}

// Workgroup barrier that waits for this wave's LDS traffic only: the key slots prefetched for the NEXT step stay in
// flight across the phase boundaries whatever the compiler would put in front of __syncthreads().  (What cost 1,150 of
// 8,170 cycles per step was not a vmcnt drain -- this barrier alone changed nothing -- but WHERE the prefetch was issued:
// see the loop, and profiles/exp/logs/r3m_latency_wide2_phases.log.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int L, bool FAST>
__global__ __launch_bounds__(64 * kWide2Waves, 1) void k_blind_rotate_wide2(BlindRotateArgs A) {
  constexpr int W = 2 * L;  // forward waves (one per decomposition row)
  constexpr int NT = 64 * kWide2Waves;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *tiles = reinterpret_cast<double2 *>(smem);                                       // [W] transpose tiles = spectra
  double2 *sums = reinterpret_cast<double2 *>(smem + (size_t)W * kTileBytes);               // [2][8][64]
  uint32_t *wbuf = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(sums) + (size_t)2 * kN2 * 16);  // [2][N]
  uint32_t *acc = wbuf + 2 * kN;                                                            // [2][N]
  double2 *t2tab = reinterpret_cast<double2 *>(reinterpret_cast<unsigned char *>(acc) + kAccBytes);
  uint16_t *s_abar = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(t2tab) + kT2Bytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t ct = blockIdx.x;
  const int n = A.n;
  const unsigned long long clk0 = A.clk ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rtc0 = A.clk ? __builtin_amdgcn_s_memrealtime() : 0ull;

  Twiddles tw;
  tw.load(A.tw, t2tab, lane);  // every wave stores the same 64 entries; ends with a workgroup barrier

  // ---- gate linear prep + rotation amounts (gates.rs:54-150, trgsw.rs:202-211) -----------------
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
    s_abar[i] = (uint16_t)((uint32_t)(p + (1u << 20)) >> 21);
  }
  uint32_t pbody = gca * pa[n];
  if (pb) pbody += gcb * pb[n];
  pbody += gcc;
  const int b_tilda = 2 * kN - (int)(((uint64_t)pbody + (1ull << 20)) >> 21);
  const uint32_t *tv = A.testvec + ct * A.per_ct_stride;
  for (int j = tid; j < kN; j += NT) {
    acc[j] = rot_read(tv, j, b_tilda);
    acc[kN + j] = rot_read(tv + kN, j, b_tilda);
  }
  __syncthreads();

  const bool fwd = wave < W;  // wave-uniform
  const int half_sel = fwd ? wave / L : 0, d = fwd ? wave % L : 0;
  const int bgbit = A.bgbit;
  const int shift = 32 - (d + 1) * bgbit;
  uint32_t signmask = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) signmask |= 1u << (32 - i * bgbit - 1);
  const uint32_t offset = A.offset;
  double2 *mytile = tiles + (size_t)(fwd ? wave : 0) * kTileCplx;
  constexpr uint32_t per_i_bytes = 2u * L * 2u * kN2 * 16u;
  const __amdgpu_buffer_rsrc_t bsk_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)A.bsk, 0, (int)((uint32_t)n * per_i_bytes), 0x00020000);
  const uint32_t lane_off = (uint32_t)lane * 16u;
  // slot `wave` of every key row (a and b halves) of the CURRENT step, loaded one step ahead
  const uint32_t my_slot = (uint32_t)wave * 1024u;
  f64x2 va[W], vb[W];
#pragma unroll
  for (int r = 0; r < W; ++r) {
    va[r] = ldkey(bsk_rsrc, lane_off, (uint32_t)r * (2u * kN2 * 16u) + my_slot);
    vb[r] = ldkey(bsk_rsrc, lane_off, (uint32_t)r * (2u * kN2 * 16u) + (uint32_t)(kN2 * 16) + my_slot);
  }
  // P0 roles: the 16 coefficient positions of a lane (m' = 0..7: j = lane + 64 m', 8..15: + 512) of each half are
  // dealt to FOUR waves -- the half's L forward waves and 4 - L of the 8 - 2L waves that have no row
  constexpr int kExtra = 4 - L;  // extra preparing waves per half
  const int prep_half = fwd ? half_sel : (wave - W) / kExtra;
  const int prep_idx = fwd ? d : L + (wave - W) % kExtra;

#pragma unroll 1
  for (int i = 0; i < n; ++i) {
    const int k = s_abar[i];
    // Next step's key slots (clamped on the last step: a harmless re-read).  96 KiB per step go through the CU's one
    // vector-memory path at 64 B per cycle = 1,536 cycles of it, and a wave is held at the instruction while the path's
    // queue is full.  Issued by all eight waves before the barrier that ends P2, they delayed the inverse transforms by
    // 900-1,150 cycles per step (measured, also with the loads hitting L1).  So the six waves that idle through P3
    // issue theirs after that barrier, and waves 0 / 1 after their inverse transform, when the path is empty again.
    auto load_next_keys = [&]() {
      const uint32_t nxt = ((TFHE_ABL_LAT & 256) ? (uint32_t)(i & 1) : (uint32_t)(i + 1 < n ? i + 1 : i)) * per_i_bytes + my_slot;
#pragma unroll
      for (int r = 0; r < ((TFHE_ABL_LAT & 64) ? 0 : W); ++r) {
        va[r] = ldkey(bsk_rsrc, lane_off, nxt + (uint32_t)r * (2u * kN2 * 16u));
        vb[r] = ldkey(bsk_rsrc, lane_off, nxt + (uint32_t)r * (2u * kN2 * 16u) + (uint32_t)(kN2 * 16));
      }
    };
    double re[8], im[8];
    {
      // P0: my quarter of w = (X^k acc - acc + offset) ^ signmask for my half (trgsw.rs:183-186 + the digit offset)
      const uint32_t *p = acc + prep_half * kN;
      uint32_t *wb = wbuf + prep_half * kN;
      if (!(TFHE_ABL_LAT & 128)) {
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
          const int j = lane + 64 * (4 * prep_idx + mm);
          wb[j] = (rot_read(p, j, k) - p[j] + offset) ^ signmask;
        }
      }
    }
    lds_barrier();
    if (fwd) {
      const uint32_t *wb = wbuf + half_sel * kN;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        re[m] = (double)sbfe(wb[lane + 64 * m], shift, bgbit);
        im[m] = (double)sbfe(wb[lane + 64 * m + kN2], shift, bgbit);
      }
      // P1: forward transform; the spectrum stays in this wave's tile, slot-major: [s][lane]
      if (!(TFHE_ABL_LAT & 8)) fft_forward(re, im, tw, mytile, lane);
      wave_lds_sync();  // the transform's last tile reads are done
#pragma unroll
      for (int s = 0; s < 8; ++s) mytile[s * 64 + lane] = make_double2(re[s], im[s]);
    }
    lds_barrier();
    if (!(TFHE_ABL_LAT & 32)) {
      // P2: slot `wave` of fa = sum_r spectrum_r * Ka_r, fb = sum_r spectrum_r * Kb_r, rows in the batch kernel's order
      double far, fai, fbr, fbi;
#pragma unroll
      for (int r = 0; r < W; ++r) {
        const double2 x = tiles[(size_t)r * kTileCplx + wave * 64 + lane];
        if (r == 0) {
          cmac<true>(far, fai, x.x, x.y, va[r]);
          cmac<true>(fbr, fbi, x.x, x.y, vb[r]);
        } else {
          cmac<false>(far, fai, x.x, x.y, va[r]);
          cmac<false>(fbr, fbi, x.x, x.y, vb[r]);
        }
      }
      sums[wave * 64 + lane] = make_double2(far, fai);
      sums[kN2 + wave * 64 + lane] = make_double2(fbr, fbi);
    }
    lds_barrier();
    if (wave < 2) {  // P3: wave 0 the a spectrum, wave 1 the b spectrum
      double f_re[8], f_im[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double2 v = sums[wave * kN2 + s * 64 + lane];
        f_re[s] = v.x;
        f_im[s] = v.y;
      }
      if (!(TFHE_ABL_LAT & 16)) fft_inverse(f_re, f_im, tw, tiles + (size_t)wave * kTileCplx, lane);
      __builtin_amdgcn_sched_barrier(0);
      load_next_keys();
      uint32_t *q = acc + wave * kN;
#pragma unroll
      for (int m = 0; m < 8; ++m) {  // res = ext + in1 (trgsw.rs:189-193)
        const int j = lane + 64 * m;
        acc_add(&q[j], round_product<FAST>(f_re[m]));
        acc_add(&q[j + kN2], round_product<FAST>(f_im[m]));
      }
    } else {
      load_next_keys();  // the six waves without an inverse transform: AFTER the barrier, so that nobody waits for the issue
    }
    lds_barrier();  // the accumulator is final for this step
  }

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

// ---------------------------------------------------------------------------------------------------------------
// k_blind_rotate_pair: the eight-wave kernel with TWO ciphertexts per workgroup, half a step apart.
//
// In k_blind_rotate_wide2 six waves idle through the inverse transforms and two through the forward ones; that is the
// price of one ciphertext's dependence chain and the right trade for up to one ciphertext per CU.  From there up to a
// few thousand ciphertexts (where the batch kernel's one wave per ciphertext still leaves SIMDs empty) a workgroup
// takes two, A and B, and runs A's forward transforms (waves 2 .. 2l+1) beside B's inverse transforms (waves 0, 1)
// and vice versa:
//     alpha    P1(A, i)  ||  P3(B, i-1)          beta     P2(A, i), then P0(B, i)
//     alpha'   P1(B, i)  ||  P3(A, i)            beta'    P2(B, i), then P0(A, i+1)
// Four barriers per TWO CMUX steps, every SIMD with two busy waves in the transform phases, and both ciphertexts
// multiply against the same key slots (loaded once per step).  Phases, arithmetic and the order of every
// floating-point operation are those of k_blind_rotate_wide2 (and so of the batch kernel): same bits.
constexpr int kPairWaves = 8;

__host__ __device__ __forceinline__ size_t blind_rotate_pair_lds_bytes(int n) {
  // 8 tiles (waves 0/1: inverse scratch; 2..: forward / spectra) | sums [2 ct][2][512] | digit words [2 ct][2][N] |
  // accumulators [2 ct][2][N] | T2 | a_bar [2 ct][n]
  return ((size_t)8 * kTileBytes + (size_t)2 * 2 * kN2 * 16 + (size_t)2 * 2 * kN * 4 + (size_t)2 * kAccBytes + kT2Bytes +
          (size_t)2 * (((size_t)n * 2 + 15) & ~(size_t)15) + 15) & ~(size_t)15;
}

template <int L, bool FAST>
__global__ __launch_bounds__(64 * kPairWaves, 1) void k_blind_rotate_pair(BlindRotateArgs A) {
  constexpr int W = 2 * L;  // forward waves: 2 .. 2 + W - 1 (row r = wave - 2)
  constexpr int NT = 64 * kPairWaves;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *tiles = reinterpret_cast<double2 *>(smem);                                            // [8]
  double2 *sums = reinterpret_cast<double2 *>(smem + (size_t)8 * kTileBytes);                    // [2 ct][2][512]
  uint32_t *wbuf = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(sums) + (size_t)2 * 2 * kN2 * 16);  // [2 ct][2][N]
  uint32_t *acc = wbuf + 2 * 2 * kN;                                                             // [2 ct][2][N]
  double2 *t2tab = reinterpret_cast<double2 *>(reinterpret_cast<unsigned char *>(acc) + 2 * kAccBytes);
  const int n = A.n;
  const size_t abar_stride = (((size_t)n * 2 + 15) & ~(size_t)15) / 2;  // u16 elements
  uint16_t *s_abar = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(t2tab) + kT2Bytes);  // [2 ct][abar_stride]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned long long clk0 = A.clk ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rtc0 = A.clk ? __builtin_amdgcn_s_memrealtime() : 0ull;

  Twiddles tw;
  tw.load(A.tw, t2tab, lane);  // ends with a workgroup barrier

  // ---- gate linear prep + rotation amounts + X^b~ testvec for both ciphertexts (gates.rs:54-150, trgsw.rs:202-211)
  size_t cts[2];
  cts[0] = (size_t)2 * blockIdx.x;
  cts[1] = cts[0] + 1 < A.count ? cts[0] + 1 : cts[0];  // an odd batch: the last workgroup runs its ciphertext twice
  const bool b_valid = cts[0] + 1 < A.count;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const size_t ct = cts[c];
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
      s_abar[c * abar_stride + i] = (uint16_t)((uint32_t)(p + (1u << 20)) >> 21);
    }
    uint32_t pbody = gca * pa[n];
    if (pb) pbody += gcb * pb[n];
    pbody += gcc;
    const int b_tilda = 2 * kN - (int)(((uint64_t)pbody + (1ull << 20)) >> 21);
    const uint32_t *tv = A.testvec + ct * A.per_ct_stride;
    uint32_t *q = acc + c * 2 * kN;
    for (int j = tid; j < kN; j += NT) {
      q[j] = rot_read(tv, j, b_tilda);
      q[kN + j] = rot_read(tv + kN, j, b_tilda);
    }
  }
  __syncthreads();

  const bool fwd = wave >= 2 && wave < 2 + W;  // wave-uniform
  const int row = fwd ? wave - 2 : 0;
  const int half_sel = row / L, d = row % L;
  const int bgbit = A.bgbit;
  const int shift = 32 - (d + 1) * bgbit;
  uint32_t signmask = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) signmask |= 1u << (32 - i * bgbit - 1);
  const uint32_t offset = A.offset;
  double2 *mytile = tiles + (size_t)wave * kTileCplx;
  constexpr uint32_t per_i_bytes = 2u * L * 2u * kN2 * 16u;
  const __amdgpu_buffer_rsrc_t bsk_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)A.bsk, 0, (int)((uint32_t)n * per_i_bytes), 0x00020000);
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t my_slot = (uint32_t)wave * 1024u;  // P2: slot `wave` of every key row, a and b halves
  f64x2 va[W], vb[W];
  auto load_keys = [&](int step) {
    const uint32_t base = (uint32_t)step * per_i_bytes + my_slot;
#pragma unroll
    for (int r = 0; r < W; ++r) {
      va[r] = ldkey(bsk_rsrc, lane_off, base + (uint32_t)r * (2u * kN2 * 16u));
      vb[r] = ldkey(bsk_rsrc, lane_off, base + (uint32_t)r * (2u * kN2 * 16u) + (uint32_t)(kN2 * 16));
    }
  };
  load_keys(0);

  // P0(c, k): w = (X^k acc - acc + offset) ^ signmask of ciphertext c, a quarter of a half per wave (all 8 waves)
  auto digit_prep = [&](int c, int k) {
    const int h = wave >> 2, idx = wave & 3;
    const uint32_t *p = acc + (c * 2 + h) * kN;
    uint32_t *wb = wbuf + (c * 2 + h) * kN;
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
      const int j = lane + 64 * (4 * idx + mm);
      wb[j] = (rot_read(p, j, k) - p[j] + offset) ^ signmask;
    }
  };
  // P1(c): forward transform of my row's digit; the spectrum stays in my tile, slot-major
  auto forward = [&](int c) {
    double re[8], im[8];
    const uint32_t *wb = wbuf + (c * 2 + half_sel) * kN;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      re[m] = (double)sbfe(wb[lane + 64 * m], shift, bgbit);
      im[m] = (double)sbfe(wb[lane + 64 * m + kN2], shift, bgbit);
    }
    if (!(TFHE_ABL_LAT & 8)) fft_forward(re, im, tw, mytile, lane);
    wave_lds_sync();
#pragma unroll
    for (int s = 0; s < 8; ++s) mytile[s * 64 + lane] = make_double2(re[s], im[s]);
  };
  // P2(c): slot `wave` of fa, fb; rows in the batch kernel's order
  auto mac = [&](int c) {
    double far, fai, fbr, fbi;
#pragma unroll
    for (int r = 0; r < W; ++r) {
      const double2 x = tiles[(size_t)(2 + r) * kTileCplx + wave * 64 + lane];
      if (r == 0) {
        cmac<true>(far, fai, x.x, x.y, va[r]);
        cmac<true>(fbr, fbi, x.x, x.y, vb[r]);
      } else {
        cmac<false>(far, fai, x.x, x.y, va[r]);
        cmac<false>(fbr, fbi, x.x, x.y, vb[r]);
      }
    }
    double2 *sm = sums + (size_t)c * 2 * kN2;
    sm[wave * 64 + lane] = make_double2(far, fai);
    sm[kN2 + wave * 64 + lane] = make_double2(fbr, fbi);
  };
  // P3(c) on waves 0 / 1: inverse transform of spectrum `wave`, rounding, accumulator update (trgsw.rs:189-193)
  auto inverse = [&](int c) {
    double f_re[8], f_im[8];
    const double2 *sm = sums + ((size_t)c * 2 + wave) * kN2;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const double2 v = sm[s * 64 + lane];
      f_re[s] = v.x;
      f_im[s] = v.y;
    }
    if (!(TFHE_ABL_LAT & 16)) fft_inverse(f_re, f_im, tw, mytile, lane);
    uint32_t *q = acc + (c * 2 + wave) * kN;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int j = lane + 64 * m;
      acc_add(&q[j], round_product<FAST>(f_re[m]));
      acc_add(&q[j + kN2], round_product<FAST>(f_im[m]));
    }
  };

  digit_prep(0, s_abar[0]);
  lds_barrier();
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
    // This step's key slots: 96 KiB through the CU's one vector-memory path.  Issued AFTER the barrier that follows
    // their registers' last use (nobody waits at a barrier for the issue, see k_blind_rotate_wide2), a transform ahead
    // of their first use in beta.
    if (i > 0 && !(TFHE_ABL_LAT & 64)) load_keys(i);
    // alpha: forward transforms of A beside the inverse transforms of B's previous step
    if (fwd) forward(0);
    else if (wave < 2 && i > 0) inverse(1);
    lds_barrier();
    // beta: multiply A's spectra; B's accumulator is final: its digits for this step
    mac(0);
    digit_prep(1, s_abar[abar_stride + i]);
    lds_barrier();
    // alpha'
    if (fwd) forward(1);
    else if (wave < 2) inverse(0);
    lds_barrier();
    // beta': multiply B's spectra (the last use of this step's key slots), A's digits for the next step
    mac(1);
    if (i + 1 < n) digit_prep(0, s_abar[i + 1]);
    lds_barrier();
  }
  if (wave < 2) inverse(1);
  lds_barrier();

#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c == 1 && !b_valid) break;
    const size_t ct = cts[c];
    const uint32_t *q = acc + c * 2 * kN;
    if (A.out_trlwe) {
      uint32_t *o = A.out_trlwe + ct * (size_t)(2 * kN);
      for (int j = tid; j < 2 * kN; j += NT) o[j] = q[j];
    }
    if (A.out_lv1) {  // trlwe.rs:106-120 with k=0
      uint32_t *o = A.out_lv1 + ct * (size_t)(kN + 1);
      for (int i = tid; i < kN; i += NT) o[i] = i == 0 ? q[0] : ~q[kN - i];
      if (tid == 0) o[kN] = q[kN];
    }
    if (A.out_ext2) {  // trlwe.rs:122-136 with k=0
      uint32_t *o = A.out_ext2 + ct * (size_t)(n + 1);
      for (int i = tid; i < n; i += NT) o[i] = i == 0 ? q[0] : ~q[n - i];
      if (tid == 0) o[n] = q[kN];
    }
  }
  if (A.clk && tid == 0) {
    atomicAdd(&A.clk[0], __builtin_amdgcn_s_memtime() - clk0);
    atomicAdd(&A.clk[1], __builtin_amdgcn_s_memrealtime() - rtc0);
  }
}


}  // namespace tfhe
