// fft512.hpp -- wave-resident negacyclic FFT over R[X]/(X^1024+1) for gfx950.
//
// One 64-lane wavefront owns one polynomial.  The 1024 real coefficients are
// folded to 512 complex points z[j] = x[j] + i*x[j+512] (the reference's
// Klemsa fold, src/fft/klemsa.rs:88-101) and transformed with three radix-8
// passes held in registers (8 complex points per lane), joined by two
// transposes through a wave-private LDS tile.
//
// No twiddle is ever applied as a separate multiply.  With j = l1 + 8*l2 + 64*m
// (lane l = l1 + 8*l2, slot m) and bin k = k1 + 8*k2 + 64*k3, the twist
// exp(i*pi*j/N) of the reference (klemsa.rs:49-58,98-100) and every inter-pass
// twiddle of the Cooley-Tukey split are GEOMETRIC in the slot index of the pass
// that consumes them:
//   pass 1 (over m  -> k1):  x_m  * c1^m ,  c1 = exp(i*pi/16)              (compile-time)
//   pass 2 (over l2 -> k2):  x_l2 * c2^l2,  c2 = exp(i*pi*(1-4*k1)/128)    (8 values, LDS table)
//   pass 3 (over l1 -> k3):  x_l1 * c3^l1,  c3 = exp(i*pi*(1-4*kap)/1024)  (kap = k1+8*k2 = per lane)
// and  sum_m x_m c^m W8^(mk)  is a DFT-8 evaluated at the points c*W8^k ("shifted DFT-8"): a
// radix-2 decimation-in-time network whose twelve butterflies (a, b) -> (a + w*b, a - w*b) carry the
// twiddles w in {c^4; c^2, -i*c^2; c, c*W8, -i*c, -i*c*W8} and cost six FMAs each
// (X = a + w*b: four, Y = 2a - X: two).  216 f64 instructions per forward transform (it was 244 with
// separate twist / twiddle multiplies) and no twiddle-table read in passes 1 and 3.
//
// Forward, lane l, slot m holds z[l+64m]:
//   pass 1 -> transpose A: (k1; l=l1+8*l2) -> lane k1*8+l1, slot l2
//   pass 2 -> transpose B: (k1,l1; k2)     -> lane k1*8+k2, slot l1
//   pass 3 -> lane mu, slot s holds bin k = (mu>>3) + 8*(mu&7) + 64*s
//           = unscaled DFT_512 of the twisted fold (the reference stores 2x that).
// The inverse walks the same index map backwards (conjugate kernels) and consumes that same bin
// order, so no bit-reversal pass exists anywhere: the bootstrapping key is permuted into this order
// once at upload.  Its twiddles are pre-twiddles of passes 2 and 3 in the same way; what cannot be
// folded is the un-twist exp(-i*pi*(l+64m)/N), which sits on the OUTPUT index: its lane part G(l)
// rides on the first butterfly stage of pass 3 (one extra complex multiply per butterfly), its slot
// part exp(-i*pi*m/16) is seven constant multiplies at the end.
//
// LDS tile: 8 planes of 72 complex (stride padded from 64 so that both
// transposes are bank-conflict free for ds_read_b128 / ds_write_b128).
#pragma once
#ifndef TFHE_FFT_HOST_EMU  // tests/cpp/test_fft_host.cpp supplies double2 / fma and empty qualifiers
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "experiment.hpp"

namespace tfhe {

constexpr int kN = 1024;
constexpr int kN2 = 512;
constexpr int kPlane = 72;                 // complex elements per k1 plane
constexpr int kTileCplx = 8 * kPlane;      // 576 complex = 9216 bytes
constexpr int kTileBytes = kTileCplx * 16;

// exp(i*pi*m/16), m = 0..7
__device__ constexpr double kCmRe[8] = {1.0,
                                        0.98078528040323044912618223613424,
                                        0.92387953251128675612818318939679,
                                        0.83146961230254523707878837761791,
                                        0.70710678118654752440084436210485,
                                        0.55557023301960222474283081394853,
                                        0.38268343236508977172845998403040,
                                        0.19509032201612826784828486847702};
__device__ constexpr double kCmIm[8] = {0.0,
                                        0.19509032201612826784828486847702,
                                        0.38268343236508977172845998403040,
                                        0.55557023301960222474283081394853,
                                        0.70710678118654752440084436210485,
                                        0.83146961230254523707878837761791,
                                        0.92387953251128675612818318939679,
                                        0.98078528040323044912618223613424};

// Twiddle table of the context (computed on the host in long double, tfhe_hip.hip make_twiddles):
//   [0   .. 255]  forward pass 3, per lane:  [q*64 + lane], q = 0..3 : c3^4, c3^2, c3, c3*W8
//                 with c3 = exp(i*pi*(1-4*kap)/1024), kap = (lane>>3) + 8*(lane&7)
//   [256 .. 575]  inverse pass 3, per lane:  [256 + q*64 + lane], q = 0..4 : G, G*c^4, c^2, c, c*conj(W8)
//                 with c = exp(2*pi*i*lane/512), G = exp(-i*pi*lane/1024)
//   [576 .. 639]  the 1 KiB LDS table: forward pass 2 [q*8 + k1] (c2^4, c2^2, c2, c2*W8) then
//                 inverse pass 2 [32 + q*8 + l1] (c^4, c^2, c, c*conj(W8) with c = exp(2*pi*i*l1/64))
// The per-lane constants (36 VGPRs) stay in registers for the whole kernel; the pass-2 constants
// have only 8 distinct values per entry and are read from LDS as they are used.
constexpr int kTwEntries = 640;
constexpr int kT2Bytes = 64 * 16;
struct Twiddles {
  double f3[8];   // forward pass 3: (re, im) x {c^4, c^2, c, c*W8}
  double i3[10];  // inverse pass 3: (re, im) x {G, G*c^4, c^2, c, c*conj(W8)}
  const double2 *t2;
  __device__ __forceinline__ void load(const double2 *__restrict__ tw, double2 *t2_lds, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double2 a = tw[q * 64 + lane];
      f3[2 * q] = a.x;
      f3[2 * q + 1] = a.y;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      double2 a = tw[256 + q * 64 + lane];
      i3[2 * q] = a.x;
      i3[2 * q + 1] = a.y;
    }
    t2_lds[lane] = tw[576 + lane];
    t2 = t2_lds;
    __syncthreads();
  }
  // The inverse-pass-3 constants again, from the (cache-resident, 10 KiB) table: a kernel that runs
  // long forward phases between its inverse transforms calls this just before them with an offset the
  // compiler cannot see through (always 0), so the 20 registers are live only across the inverse
  // transforms instead of the whole kernel.
  __device__ __forceinline__ void reload_i3(const double2 *__restrict__ tw, int lane, int opaque_zero) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      double2 a = tw[256 + q * 64 + lane + opaque_zero];
      i3[2 * q] = a.x;
      i3[2 * q + 1] = a.y;
    }
  }
};

// Plain in-register 8-point DFT (no pre-twiddle).  INV=false: W8 = exp(-2*pi*i/8); INV=true: conjugate.
// The two 1/sqrt2 twiddles are not applied where they arise: their common factor H is
// carried to the last stage and folded into its add/sub as FMAs (X = b +- H*q).  52 instructions.
// (TFHE_ABL_NOFFT: timing-only experiment switch, see experiment.hpp -- 0 in every product build)
template <bool INV>
__device__ __forceinline__ void dft8(double (&re)[8], double (&im)[8]) {
  if (TFHE_ABL_NOFFT) return;
  constexpr double H = 0.70710678118654752440084436210485;
  double a0r = re[0] + re[4], a0i = im[0] + im[4];
  double a4r = re[0] - re[4], a4i = im[0] - im[4];
  double a1r = re[1] + re[5], a1i = im[1] + im[5];
  double t5r = re[1] - re[5], t5i = im[1] - im[5];
  double a2r = re[2] + re[6], a2i = im[2] + im[6];
  double t6r = re[2] - re[6], t6i = im[2] - im[6];
  double a3r = re[3] + re[7], a3i = im[3] + im[7];
  double t7r = re[3] - re[7], t7i = im[3] - im[7];
  // p5 = t5 * (1 -+ i), p7 = t7 * (-1 -+ i)   (a5 = H*p5, a7 = H*p7), a6 = t6 * (-+i)
  double p5r, p5i, a6r, a6i, p7r, p7i;
  if (!INV) {
    p5r = t5r + t5i;  p5i = t5i - t5r;
    a6r = t6i;        a6i = -t6r;
    p7r = t7i - t7r;  p7i = -(t7r + t7i);
  } else {
    p5r = t5r - t5i;  p5i = t5r + t5i;
    a6r = -t6i;       a6i = t6r;
    p7r = -(t7r + t7i);  p7i = t7r - t7i;
  }
  // even half
  double b0r = a0r + a2r, b0i = a0i + a2i;
  double b2r = a0r - a2r, b2i = a0i - a2i;
  double b1r = a1r + a3r, b1i = a1i + a3i;
  double u3r = a1r - a3r, u3i = a1i - a3i;
  double b3r, b3i;
  if (!INV) { b3r = u3i; b3i = -u3r; } else { b3r = -u3i; b3i = u3r; }
  // odd half: b5 = H*q5 with q5 = p5 + p7; b7 = H*q7 with q7 = (p5 - p7) * (-+i)
  double b4r = a4r + a6r, b4i = a4i + a6i;
  double b6r = a4r - a6r, b6i = a4i - a6i;
  double q5r = p5r + p7r, q5i = p5i + p7i;
  double u7r = p5r - p7r, u7i = p5i - p7i;
  double q7r, q7i;
  if (!INV) { q7r = u7i; q7i = -u7r; } else { q7r = -u7i; q7i = u7r; }
  re[0] = b0r + b1r; im[0] = b0i + b1i;
  re[4] = b0r - b1r; im[4] = b0i - b1i;
  re[2] = b2r + b3r; im[2] = b2i + b3i;
  re[6] = b2r - b3r; im[6] = b2i - b3i;
  re[1] = fma(H, q5r, b4r);  im[1] = fma(H, q5i, b4i);
  re[5] = fma(-H, q5r, b4r); im[5] = fma(-H, q5i, b4i);
  re[3] = fma(H, q7r, b6r);  im[3] = fma(H, q7i, b6i);
  re[7] = fma(-H, q7r, b6r); im[7] = fma(-H, q7i, b6i);
}

// x *= (wr + i*wi)  or, CONJ, x *= (wr - i*wi)
template <bool CONJ>
__device__ __forceinline__ void cmul(double &xr, double &xi, double wr, double wi) {
  double r, i;
  if (!CONJ) {
    r = xr * wr - xi * wi;
    i = xr * wi + xi * wr;
  } else {
    r = xr * wr + xi * wi;
    i = xi * wr - xr * wi;
  }
  xr = r;
  xi = i;
}

// Radix-2 decimation-in-time butterfly with its twiddle folded in: (a, b) <- (a + w*b, a - w*b).
// X = a + w*b is two chained FMAs per component, Y = 2a - X one: six instructions, twiddle included.
__device__ __forceinline__ void bfly(double &ar, double &ai, double &br, double &bi, double wr, double wi) {
  const double xr = fma(-wi, bi, fma(wr, br, ar));
  const double xi = fma(wi, br, fma(wr, bi, ai));
  br = fma(2.0, ar, -xr);
  bi = fma(2.0, ai, -xi);
  ar = xr;
  ai = xi;
}

// Shifted DFT-8:  X_k = sum_m x_m * (c * W8^k)^m,  W8 = exp(-2*pi*i/8) (INV: conjugate),
// from the four constants c^4, c^2, c, c*W8 (INV: c*conj(W8)).  With SCALE the inputs are first
// multiplied by g: the caller passes g and g*c^4 (the factor only has to reach the `a` operand of the
// first stage; the `b` operand gets it through the twiddle).  72 instructions (88 with SCALE).
template <bool INV, bool SCALE = false>
__device__ __forceinline__ void sdft8(double (&re)[8], double (&im)[8], double c4r, double c4i, double c2r,
                                      double c2i, double c1r, double c1i, double cwr, double cwi,
                                      double gr = 1.0, double gi = 0.0) {
  if (TFHE_ABL_NOFFT) return;
  if (SCALE) {
    cmul<false>(re[0], im[0], gr, gi);
    cmul<false>(re[2], im[2], gr, gi);
    cmul<false>(re[1], im[1], gr, gi);
    cmul<false>(re[3], im[3], gr, gi);
  }
  // stage 1: (x0,x4) (x2,x6) (x1,x5) (x3,x7), twiddle c^4
  bfly(re[0], im[0], re[4], im[4], c4r, c4i);
  bfly(re[2], im[2], re[6], im[6], c4r, c4i);
  bfly(re[1], im[1], re[5], im[5], c4r, c4i);
  bfly(re[3], im[3], re[7], im[7], c4r, c4i);
  // stage 2: twiddles c^2 and (-+i)*c^2      (-i*w = (wi, -wr);  +i*w = (-wi, wr))
  bfly(re[0], im[0], re[2], im[2], c2r, c2i);
  bfly(re[1], im[1], re[3], im[3], c2r, c2i);
  if (!INV) {
    bfly(re[4], im[4], re[6], im[6], c2i, -c2r);
    bfly(re[5], im[5], re[7], im[7], c2i, -c2r);
  } else {
    bfly(re[4], im[4], re[6], im[6], -c2i, c2r);
    bfly(re[5], im[5], re[7], im[7], -c2i, c2r);
  }
  // stage 3: twiddles c, c*W8, (-+i)*c, (-+i)*c*W8
  bfly(re[0], im[0], re[1], im[1], c1r, c1i);
  bfly(re[4], im[4], re[5], im[5], cwr, cwi);
  if (!INV) {
    bfly(re[2], im[2], re[3], im[3], c1i, -c1r);
    bfly(re[6], im[6], re[7], im[7], cwi, -cwr);
  } else {
    bfly(re[2], im[2], re[3], im[3], -c1i, c1r);
    bfly(re[6], im[6], re[7], im[7], -cwi, cwr);
  }
  // registers now hold X0 X4 X2 X6 X1 X5 X3 X7 -> natural order (a renaming, no instructions)
  const double r1 = re[4], i1 = im[4], r3 = re[6], i3 = im[6], r4 = re[1], i4 = im[1], r6 = re[3], i6 = im[3];
  re[1] = r1; im[1] = i1;
  re[3] = r3; im[3] = i3;
  re[4] = r4; im[4] = i4;
  re[6] = r6; im[6] = i6;
}

// LDS hand-off WITHIN one wavefront: orders this wave's LDS writes before its later LDS reads of
// other lanes' data.  LDS operations of a wave execute in program order, so the hardware needs no
// barrier; what is needed is that the compiler keeps the order (memory clobber) and that pending
// reads have returned before their registers are reused (lgkmcnt).  It involves no other wave, so
// the same FFT code serves one-wave workgroups and the multi-wave latency kernel.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Compiler-only ordering point for LDS traffic: the hardware keeps a wave's LDS operations in program
// order, so a write may follow the reads of the same tile without a wait; the compiler's own
// s_waitcnt protects the registers.
__device__ __forceinline__ void wave_lds_order() { asm volatile("" ::: "memory"); }
// compile-time constants of pass 1: c1 = exp(i*pi/16): c1^4 = exp(i*pi/4), c1^2 = exp(i*pi/8),
// c1*W8 = exp(-3*i*pi/16)
constexpr double kP1c4 = 0.70710678118654752440084436210485;

// Timing-only experiment switches (experiment.hpp; 0 in every product build): TFHE_ABL_NOLDS removes the
// transposes; TFHE_ABL_TPB_DPP replaces the LDS round trip of transpose B (the exchange inside 8-lane groups) by
// the instruction mix a cross-lane version would issue -- lane bit 5 by 16 v_permlane32_swap, lane bit 4 by 16
// v_permlane16_swap, lane bit 3 by 48 DPP moves.  It answered "would DPP / permlane transposes be faster than
// LDS ones?" without building the re-indexed FFT (no: 331.5 vs 329.9 ms).
#if !defined(TFHE_FFT_HOST_EMU) && TFHE_ABL_TPB_DPP
__device__ __forceinline__ void abl_crosslane_exchange(double (&re)[8], double (&im)[8]) {
  uint32_t w[32];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    w[4 * i] = (uint32_t)__double2loint(re[i]);
    w[4 * i + 1] = (uint32_t)__double2hiint(re[i]);
    w[4 * i + 2] = (uint32_t)__double2loint(im[i]);
    w[4 * i + 3] = (uint32_t)__double2hiint(im[i]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(w[i]), "+v"(w[i + 16]));
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(w[2 * i]), "+v"(w[2 * i + 1]));
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    uint32_t t;
    asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "=v"(t) : "v"(w[i]), "0"(w[i + 16]));
    asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(w[i]) : "v"(w[i + 16]));
    asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3" : "+v"(w[i + 16]) : "v"(t));
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    re[i] = __hiloint2double((int)w[4 * i + 1], (int)w[4 * i]);
    im[i] = __hiloint2double((int)w[4 * i + 3], (int)w[4 * i + 2]);
  }
}
#endif

// ---- the two transposes, as separable halves so that two transforms can be interleaved ---------
// A: write (k1, l) at k1*72 + l ; read lane (k1', l1) slot l2 at k1'*72 + l1 + 8*l2
// B: write (k1, k2, l1) at k1*72 + k2*9 + l1 ; read lane (k1, k2') slot l1 at k1*72 + k2'*9 + l1
__device__ __forceinline__ void tpA_write(const double (&re)[8], const double (&im)[8], double2 *tile, int lane) {
  if (TFHE_ABL_NOLDS) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) tile[k * kPlane + lane] = make_double2(re[k], im[k]);
}
__device__ __forceinline__ void tpA_read(double (&re)[8], double (&im)[8], const double2 *tile, int lane) {
  if (TFHE_ABL_NOLDS) return;
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    double2 v = tile[hi * kPlane + lo + 8 * s];
    re[s] = v.x;
    im[s] = v.y;
  }
}
__device__ __forceinline__ void tpB_write(const double (&re)[8], const double (&im)[8], double2 *tile, int lane) {
  if (TFHE_ABL_NOLDS || TFHE_ABL_TPB_DPP) return;
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int k = 0; k < 8; ++k) tile[hi * kPlane + k * 9 + lo] = make_double2(re[k], im[k]);
}
__device__ __forceinline__ void tpB_read(double (&re)[8], double (&im)[8], const double2 *tile, int lane) {
  if (TFHE_ABL_NOLDS) return;
#if !defined(TFHE_FFT_HOST_EMU) && TFHE_ABL_TPB_DPP
  abl_crosslane_exchange(re, im);
  return;
#endif
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    double2 v = tile[hi * kPlane + lo * 9 + s];
    re[s] = v.x;
    im[s] = v.y;
  }
}
// the inverse transposes are the same maps with the roles of write and read exchanged
__device__ __forceinline__ void tpBi_write(const double (&re)[8], const double (&im)[8], double2 *tile, int lane) {
  if (TFHE_ABL_NOLDS || TFHE_ABL_TPB_DPP) return;
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int s = 0; s < 8; ++s) tile[hi * kPlane + lo * 9 + s] = make_double2(re[s], im[s]);
}
__device__ __forceinline__ void tpBi_read(double (&re)[8], double (&im)[8], const double2 *tile, int lane) {
  if (TFHE_ABL_NOLDS) return;
#if !defined(TFHE_FFT_HOST_EMU) && TFHE_ABL_TPB_DPP
  abl_crosslane_exchange(re, im);
  return;
#endif
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double2 v = tile[hi * kPlane + k * 9 + lo];
    re[k] = v.x;
    im[k] = v.y;
  }
}
__device__ __forceinline__ void tpAi_write(const double (&re)[8], const double (&im)[8], double2 *tile, int lane) {
  if (TFHE_ABL_NOLDS) return;
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int s = 0; s < 8; ++s) tile[hi * kPlane + lo + 8 * s] = make_double2(re[s], im[s]);
}
__device__ __forceinline__ void tpAi_read(double (&re)[8], double (&im)[8], const double2 *tile, int lane) {
  if (TFHE_ABL_NOLDS) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double2 v = tile[k * kPlane + lane];
    re[k] = v.x;
    im[k] = v.y;
  }
}

// ---- the passes --------------------------------------------------------------------------------
__device__ __forceinline__ void fwd_pass1(double (&re)[8], double (&im)[8]) {
  sdft8<false>(re, im, kP1c4, kP1c4, kCmRe[2], kCmIm[2], kCmRe[1], kCmIm[1], kCmRe[3], -kCmIm[3]);
}
__device__ __forceinline__ void fwd_pass2(double (&re)[8], double (&im)[8], const Twiddles &tw, int lane) {
  const int hi = lane >> 3;
  const double2 c4 = tw.t2[hi], c2 = tw.t2[8 + hi], c1 = tw.t2[16 + hi], cw = tw.t2[24 + hi];
  sdft8<false>(re, im, c4.x, c4.y, c2.x, c2.y, c1.x, c1.y, cw.x, cw.y);
}
__device__ __forceinline__ void fwd_pass3(double (&re)[8], double (&im)[8], const Twiddles &tw) {
  sdft8<false>(re, im, tw.f3[0], tw.f3[1], tw.f3[2], tw.f3[3], tw.f3[4], tw.f3[5], tw.f3[6], tw.f3[7]);
}
__device__ __forceinline__ void inv_pass2(double (&re)[8], double (&im)[8], const Twiddles &tw, int lane) {
  const int lo = lane & 7;
  const double2 c4 = tw.t2[32 + lo], c2 = tw.t2[40 + lo], c1 = tw.t2[48 + lo], cw = tw.t2[56 + lo];
  sdft8<true>(re, im, c4.x, c4.y, c2.x, c2.y, c1.x, c1.y, cw.x, cw.y);
}
__device__ __forceinline__ void inv_pass3(double (&re)[8], double (&im)[8], const Twiddles &tw) {
  sdft8<true, true>(re, im, tw.i3[2], tw.i3[3], tw.i3[4], tw.i3[5], tw.i3[6], tw.i3[7], tw.i3[8], tw.i3[9],
                    tw.i3[0], tw.i3[1]);
#pragma unroll
  for (int m = 1; m < 8; ++m) cmul<true>(re[m], im[m], kCmRe[m], kCmIm[m]);
}

// Forward transform.  In: re/im[m] = fold of coefficients (l+64m, l+64m+512),
// NOT yet twisted.  Out: re/im[s] = bin (mu>>3)+8*(mu&7)+64*s, unscaled.
__device__ __forceinline__ void fft_forward(double (&re)[8], double (&im)[8], const Twiddles &tw,
                                            double2 *tile, int lane) {
  fwd_pass1(re, im);
  wave_lds_order();  // previous readers of the tile were issued earlier: LDS executes in order
  tpA_write(re, im, tile, lane);
  wave_lds_order();
  tpA_read(re, im, tile, lane);
  fwd_pass2(re, im, tw, lane);
  wave_lds_order();
  tpB_write(re, im, tile, lane);
  wave_lds_order();
  tpB_read(re, im, tile, lane);
  fwd_pass3(re, im, tw);
}

// Inverse transform (mirror).  In: bins in the forward output order.
// Out: re/im[m] = untwisted z[l+64m]: re -> coefficient l+64m, im -> l+64m+512.
// Un-normalised: the 1/1024 of the reference (0.5 in klemsa.rs:126 times
// 1/512 in :136) is folded into the operands by the caller.
__device__ __forceinline__ void fft_inverse(double (&re)[8], double (&im)[8], const Twiddles &tw,
                                            double2 *tile, int lane) {
  dft8<true>(re, im);  // over k3 -> l1
  wave_lds_order();
  tpBi_write(re, im, tile, lane);
  wave_lds_order();
  tpBi_read(re, im, tile, lane);
  inv_pass2(re, im, tw, lane);  // over k2 -> l2
  wave_lds_order();
  tpAi_write(re, im, tile, lane);
  wave_lds_order();
  tpAi_read(re, im, tile, lane);
  inv_pass3(re, im, tw);  // over k1 -> m, un-twist
}

// Two independent inverse transforms interleaved through ONE tile: while the transpose of one is on
// its way through the LDS the other's butterflies issue, so a wave covers its own LDS round trips.
// Legal on one tile because a wave's LDS operations execute in program order: y's writes cannot
// overtake x's reads.
__device__ __forceinline__ void fft_inverse2(double (&xr)[8], double (&xi)[8], double (&yr)[8], double (&yi)[8],
                                             const Twiddles &tw, double2 *tile, int lane) {
  dft8<true>(xr, xi);
  wave_lds_order();
  tpBi_write(xr, xi, tile, lane);
  wave_lds_order();
  tpBi_read(xr, xi, tile, lane);
  dft8<true>(yr, yi);  // covers x's round trip
  wave_lds_order();
  tpBi_write(yr, yi, tile, lane);
  wave_lds_order();
  tpBi_read(yr, yi, tile, lane);
  inv_pass2(xr, xi, tw, lane);  // covers y's
  wave_lds_order();
  tpAi_write(xr, xi, tile, lane);
  wave_lds_order();
  tpAi_read(xr, xi, tile, lane);
  inv_pass2(yr, yi, tw, lane);
  wave_lds_order();
  tpAi_write(yr, yi, tile, lane);
  wave_lds_order();
  tpAi_read(yr, yi, tile, lane);
  inv_pass3(xr, xi, tw);
  inv_pass3(yr, yi, tw);
}

#ifndef TFHE_FFT_HOST_EMU
// `as i64 as u32` of f64::round (klemsa.rs:145-146): low 32 bits of the rounded integer.
//
// FAST: valid when |x| < 2^51 is guaranteed (the host checks
// 2l * N * (Bg/2) * 2^31 < 2^51, true for l=3,bgbit=6): adding 1.5*2^52 leaves
// round-to-nearest(x) in the low mantissa bits and 1.5*2^52 = 0 mod 2^32.  It
// differs from f64::round only on exact .5 ties, which cannot occur where the
// FFT product is exact (|x - integer| <= 0.004).
template <bool FAST = false>
__device__ __forceinline__ uint32_t round_to_torus(double x) {
  if (FAST) {
    return (uint32_t)__double2loint(x + 0x1.8p52);
  } else {
    // |x| < 2^63: peel off the multiple of 2^32 first (exact: power-of-two scaling, round-to-integer,
    // one fused multiply-add whose result |v| <= 2^31 is representable), then the same trick on v.
    // Four instructions instead of ten for round / floor / fma / convert.  Like FAST it resolves exact
    // .5 ties to even instead of away from zero; this path serves bgbit > 10, where the f64 product is
    // ~2^7 LSB away from the integer product anyway and only phases / messages are comparable.
    const double q = rint(x * 0x1p-32);
    const double v = fma(q, -0x1p32, x);
    return (uint32_t)__double2loint(v + 0x1.8p52);
  }
}

// The rounding the blind-rotation kernels apply to the external product (round_product<FAST>) and the scale of the
// engine's bootstrapping key that goes with it (key_scale(fast)).
//   FAST sets: key x 2^-10 (the reference's 0.5 * 1/512, klemsa.rs:126,136), one add (round_to_torus<true>).
//   Other sets (bgbit > 10, |x| up to 2^63): the key carries a further 2^-32 -- a power of two, so every product, sum
//   and rounding of the transform is the same mantissa with the exponent lowered by 32 -- and the inverse transform
//   delivers y = x * 2^-32 exactly.  Then q = rint(y) is the multiple of 2^32 to drop, t = y - q (exact, |t| <= 1/2)
//   is v * 2^-32 for the same v = x - q * 2^32 as round_to_torus<false> forms, and t + 1.5 * 2^20 has its last
//   mantissa bit at 2^-32: the low word is round-to-nearest-even(v) mod 2^32 -- the same bits as
//   round_to_torus<false>(x), in three instructions instead of four (32 fewer per lane and CMUX step at l = 1).
#ifndef TFHE_ROUND_SCALED  // (build knob: 0 = the four-instruction form on a 2^-10 key, for the A/B)
#define TFHE_ROUND_SCALED 1
#endif
__host__ __device__ constexpr double key_scale(bool fast) { return (fast || !TFHE_ROUND_SCALED) ? 0x1p-10 : 0x1p-42; }
template <bool FAST>
__device__ __forceinline__ uint32_t round_product(double y) {
  if (FAST || !TFHE_ROUND_SCALED) return round_to_torus<FAST>(y);
  const double q = rint(y);
  const double t = y - q;
#if TFHE_ABL_ROUND_LSB  // mutation build (tests/test_gpu_parity.py::test_rounding_mutation_is_caught): one LSB off on ~1/1024 of the words
  const uint32_t r = (uint32_t)__double2loint(t + 0x1.8p20);
  return r + ((r & 0x3FFu) == 0x155u ? 1u : 0u);
#else
  return (uint32_t)__double2loint(t + 0x1.8p20);
#endif
}

#endif  // TFHE_FFT_HOST_EMU

// f64::round exactly as the reference's FFTProcessor::fft does it (klemsa.rs:145-146): half away
// from zero, then `as i64 as u32`.  Used by the stage entry points (tfhe_hip_batch_fft / _poly_mul),
// whose caller-supplied spectra may land on exact .5 ties; valid for |x| < 2^63.
__device__ __forceinline__ uint32_t round_half_away_to_torus(double x) {
  const double q = rint(x * 0x1p-32);   // multiple of 2^32 to drop (its own ties are irrelevant: |v| <= 2^31 either way)
  const double v = fma(q, -0x1p32, x);  // exact: x - q*2^32
  double r = rint(v);                   // ties to even ...
  if (fabs(v - r) == 0.5) r = v + copysign(0.5, x);  // ... moved away from zero (the sign is x's, not v's)
  return (uint32_t)(long long)r;
}

// bin held by (lane mu, slot s) after fft_forward
__host__ __device__ __forceinline__ int bin_of(int mu, int s) {
  return (mu >> 3) + 8 * (mu & 7) + 64 * s;
}

}  // namespace tfhe
