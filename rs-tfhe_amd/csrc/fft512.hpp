// fft512.hpp -- wave-resident negacyclic FFT over R[X]/(X^1024+1) for gfx950.
//
// One 64-lane wavefront owns one polynomial.  The 1024 real coefficients are
// folded to 512 complex points z[j] = x[j] + i*x[j+512] (the reference's
// Klemsa fold, src/fft/klemsa.rs:88-101) and transformed with three radix-8
// passes held in registers (8 complex points per lane), joined by two
// transposes through a wave-private LDS tile.  The twist exp(i*pi*j/N) of the
// reference (klemsa.rs:49-58,98-100) is split as
//     exp(i*pi*(l+64m)/N) = exp(i*pi*l/N) * exp(i*pi*m/16)
// -- the m part is a compile-time constant per register slot, the lane part is
// merged into the first pass' twiddle, so the twist costs no extra pass.
//
// Forward (decimation in frequency), lane l, slot m holds z[l+64m]:
//   pass 1: DFT-8 over m            -> k1, times T1[l][k1] = exp(i*pi*l*(1-4k1)/N)
//   transpose A: (k1; l=l1+8*l2)    -> lane k1*8+l1, slot l2
//   pass 2: DFT-8 over l2           -> k2, times T2[l1][k2] = exp(-2*pi*i*l1*k2/64)
//   transpose B: (k1,l1; k2)        -> lane k1*8+k2, slot l1
//   pass 3: DFT-8 over l1           -> k3
//   result: lane mu, slot s holds bin k = (mu>>3) + 8*(mu&7) + 64*s
//           = unscaled DFT_512 of the twisted fold (the reference stores 2x that).
// The inverse is the exact mirror (decimation in time) and consumes that same
// bin order, so no bit-reversal pass exists anywhere: the bootstrapping key is
// permuted into this order once at upload.
//
// LDS tile: 8 planes of 72 complex (stride padded from 64 so that both
// transposes are bank-conflict free for ds_read_b128 / ds_write_b128).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// Per-lane twiddles, loaded once per kernel from the context's table
// (computed on the host in long double): tw[0..511] = T1[k1][lane] as
// [k1*64+lane], tw[512..575] = T2[k2][l1] as [512 + k2*8 + l1].
//
// T1 (lane-dependent, 8 complex) stays in registers for the whole kernel; T2 has only
// 64 distinct values, so it sits in a 1 KiB LDS table read as it is used -- that frees
// 28 VGPRs, which is what lets the key-row prefetch fit beside two waves per SIMD.
constexpr int kT2Bytes = 64 * 16;
struct Twiddles {
  double t1re[8], t1im[8];
  const double2 *t2;  // LDS: [k2*8 + l1]
  __device__ __forceinline__ void load(const double2 *__restrict__ tw, double2 *t2_lds, int lane) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double2 a = tw[k * 64 + lane];
      t1re[k] = a.x;
      t1im[k] = a.y;
    }
    t2_lds[lane] = tw[512 + lane];
    t2 = t2_lds;
    __syncthreads();
  }
};

// In-register 8-point DFT.  INV=false: W8 = exp(-2*pi*i/8); INV=true: conjugate.
// The two 1/sqrt2 twiddles are not applied where they arise: their common factor H is
// carried to the last stage and folded into its add/sub as FMAs (X = b +- H*q), which
// removes the four H multiplies per butterfly.
template <bool INV>
__device__ __forceinline__ void dft8(double (&re)[8], double (&im)[8]) {
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

// LDS hand-off WITHIN one wavefront: orders this wave's LDS writes before its later LDS reads of
// other lanes' data.  LDS operations of a wave execute in program order, so the hardware needs no
// barrier; what is needed is that the compiler keeps the order (memory clobber) and that pending
// reads have returned before their registers are reused (lgkmcnt).  It involves no other wave, so
// the same FFT code serves one-wave workgroups and the multi-wave latency kernel.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Forward transform.  In: re/im[m] = fold of coefficients (l+64m, l+64m+512),
// NOT yet twisted.  Out: re/im[s] = bin (mu>>3)+8*(mu&7)+64*s, unscaled.
__device__ __forceinline__ void fft_forward(double (&re)[8], double (&im)[8], const Twiddles &tw,
                                            double2 *tile, int lane) {
#pragma unroll
  for (int m = 1; m < 8; ++m) cmul<false>(re[m], im[m], kCmRe[m], kCmIm[m]);
  dft8<false>(re, im);
#pragma unroll
  for (int k = 0; k < 8; ++k) cmul<false>(re[k], im[k], tw.t1re[k], tw.t1im[k]);
  // transpose A: write (k1, l) at k1*72 + l ; read lane (k1', l1) slot l2 at k1'*72 + l1 + 8*l2
  wave_lds_sync();  // previous readers of the tile are done
#pragma unroll
  for (int k = 0; k < 8; ++k) tile[k * kPlane + lane] = make_double2(re[k], im[k]);
  wave_lds_sync();
  const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    double2 v = tile[hi * kPlane + lo + 8 * s];
    re[s] = v.x;
    im[s] = v.y;
  }
  dft8<false>(re, im);
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const double2 w = tw.t2[k * 8 + lo];
    cmul<false>(re[k], im[k], w.x, w.y);
  }
  // transpose B: write (k1, k2, l1) at k1*72 + k2*9 + l1 ; read lane (k1, k2') slot l1
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) tile[hi * kPlane + k * 9 + lo] = make_double2(re[k], im[k]);
  wave_lds_sync();
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    double2 v = tile[hi * kPlane + lo * 9 + s];
    re[s] = v.x;
    im[s] = v.y;
  }
  dft8<false>(re, im);
}

// Inverse transform (mirror).  In: bins in the forward output order.
// Out: re/im[m] = untwisted z[l+64m]: re -> coefficient l+64m, im -> l+64m+512.
// Un-normalised: the 1/1024 of the reference (0.5 in klemsa.rs:126 times
// 1/512 in :136) is folded into the operands by the caller.
__device__ __forceinline__ void fft_inverse(double (&re)[8], double (&im)[8], const Twiddles &tw,
                                            double2 *tile, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
  dft8<true>(re, im);  // over k3 -> l1
  wave_lds_sync();
#pragma unroll
  for (int s = 0; s < 8; ++s) tile[hi * kPlane + lo * 9 + s] = make_double2(re[s], im[s]);
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double2 v = tile[hi * kPlane + k * 9 + lo];
    re[k] = v.x;
    im[k] = v.y;
  }
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const double2 w = tw.t2[k * 8 + lo];
    cmul<true>(re[k], im[k], w.x, w.y);
  }
  dft8<true>(re, im);  // over k2 -> l2
  wave_lds_sync();
#pragma unroll
  for (int s = 0; s < 8; ++s) tile[hi * kPlane + lo + 8 * s] = make_double2(re[s], im[s]);
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double2 v = tile[k * kPlane + lane];
    re[k] = v.x;
    im[k] = v.y;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) cmul<true>(re[k], im[k], tw.t1re[k], tw.t1im[k]);
  dft8<true>(re, im);  // over k1 -> m
#pragma unroll
  for (int m = 1; m < 8; ++m) cmul<true>(re[m], im[m], kCmRe[m], kCmIm[m]);
}

// f64::round (half away from zero) then `as i64 as u32` (klemsa.rs:145-146):
// low 32 bits of the rounded integer, exact for |x| < 2^63.
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

// bin held by (lane mu, slot s) after fft_forward
__host__ __device__ __forceinline__ int bin_of(int mu, int s) {
  return (mu >> 3) + 8 * (mu & 7) + 64 * s;
}

}  // namespace tfhe
