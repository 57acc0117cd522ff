// keygen.hpp -- cloud-key generation on the GPU.
//
// Replaces CloudKey::new(&secret_key) (src/key.rs:59-66) for callers that hold the secret key
// on the host: gen_key_switching_key (key.rs:102-122, the reference's slowest step: a sequential
// loop of N*t*(base-1) TLWE encryptions) and gen_bootstrapping_key (key.rs:124-156: n TRGSW
// encryptions = n*2l TRLWE zero-encryptions, each with one negacyclic product a (*) s1, then 4l
// forward FFTs).  The keys are written straight into the engine layouts, so the 172 MB upload
// and the two conversion kernels disappear.
//
// Randomness: the reference draws from an OS-seeded ChaCha thread_rng (tlwe.rs:38, trlwe.rs:36-41).  Here every
// mask word and every Gaussian sample is a fixed position of a ChaCha20 keystream (RFC 8439 block function,
// 20 rounds) under a 256-bit key: block (counter, nonce = {row, stream, domain}).  The published key rows are
// (mask, <mask, s> + noise); whoever can regenerate the noise reads the secret key off them, so the 256-bit key
// must come from the OS (tfhe_hip_gen_cloud_key_secure) or from the caller's own CSPRNG (..._with_key).  The
// 64-bit-seed entry point expands the seed into a key and exists for reproducible tests and benchmarks only.
// Distributions are the reference's: uniform u32 mask, N(0, alpha) noise added on the torus (utils.rs:9-38).
#pragma once
#include "blind_rotate.hpp"
#include "key_switch.hpp"

namespace tfhe {

// ---- ChaCha20 block function (RFC 8439 section 2.3) -------------------------------------
struct ChaChaKey {
  uint32_t k[8];
};
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
#define TFHE_QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); \
  c += d; b ^= c; b = rotl32(b, 12); \
  a += b; d ^= a; d = rotl32(d, 8);  \
  c += d; b ^= c; b = rotl32(b, 7);
// 16 keystream words of block `counter` under nonce (n0, n1, n2)
__device__ __forceinline__ void chacha20_block(const ChaChaKey &key, uint32_t counter, uint32_t n0, uint32_t n1, uint32_t n2,
                                               uint32_t (&out)[16]) {
  const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                           key.k[4],    key.k[5],    key.k[6],    key.k[7],    counter,  n0,       n1,       n2};
  uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7], x8 = in[8],
           x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
#pragma unroll 1
  for (int r = 0; r < 10; ++r) {  // 10 double rounds
    TFHE_QR(x0, x4, x8, x12)
    TFHE_QR(x1, x5, x9, x13)
    TFHE_QR(x2, x6, x10, x14)
    TFHE_QR(x3, x7, x11, x15)
    TFHE_QR(x0, x5, x10, x15)
    TFHE_QR(x1, x6, x11, x12)
    TFHE_QR(x2, x7, x8, x13)
    TFHE_QR(x3, x4, x9, x14)
  }
  out[0] = x0 + in[0];    out[1] = x1 + in[1];    out[2] = x2 + in[2];    out[3] = x3 + in[3];
  out[4] = x4 + in[4];    out[5] = x5 + in[5];    out[6] = x6 + in[6];    out[7] = x7 + in[7];
  out[8] = x8 + in[8];    out[9] = x9 + in[9];    out[10] = x10 + in[10]; out[11] = x11 + in[11];
  out[12] = x12 + in[12]; out[13] = x13 + in[13]; out[14] = x14 + in[14]; out[15] = x15 + in[15];
}
#undef TFHE_QR

// src/utils.rs:9-12
__device__ __forceinline__ uint32_t dev_f64_to_torus(double d) {
  double t = fmod(d, 1.0) * 4294967296.0;
  return (uint32_t)(long long)t;
}

// two N(0, sigma) samples from four uniform words (Box-Muller)
__device__ __forceinline__ void gauss2(const uint32_t *w, double sigma, double &g0, double &g1) {
  const double u1 = ((double)(((uint64_t)w[0] << 21) ^ (w[1] >> 11)) + 1.0) * (1.0 / 9007199254740992.0);  // (0,1]
  const double u2 = (double)(((uint64_t)w[2] << 21) ^ (w[3] >> 11)) * (1.0 / 9007199254740992.0);          // [0,1)
  const double rad = sqrt(-2.0 * log(u1)) * sigma;
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  g0 = rad * c;
  g1 = rad * s;
}

// utils.rs:22-38 gaussian_f64(mu): f64_to_torus(sample) + f64_to_torus(mu)
__device__ __forceinline__ uint32_t gaussian_torus(double mu, double g) { return dev_f64_to_torus(g) + dev_f64_to_torus(mu); }

// ---- key-switching key: key.rs:102-122, rows written in the engine layout -------------
// One workgroup per row (i, j, k); row = TLWELv0::encrypt_f64(k*s1[i] / 2^((j+1)*basebit), alpha, s0)
// (tlwe.rs:37-53): a uniform, b = <a, s0> + gaussian_f64(p).
__global__ __launch_bounds__(256) void k_gen_ksk(const uint32_t *__restrict__ key_lv0, const uint32_t *__restrict__ key_lv1,
                                                  uint32_t *__restrict__ ksk_eng, int n, int basebit, int t,
                                                  double alpha, const ChaChaKey *__restrict__ key_p) {
  __shared__ uint32_t s_part[4];
  const ChaChaKey key = *key_p;  // the generator key lives in a device buffer the host wipes after the launch
  const uint32_t row = blockIdx.x;  // base*t*i + base*j + k
  const int base = 1 << basebit;
  const int k = row % base, j = (row / base) % t, i = row / (base * t);
  const int rw = ksk_row_words(n);
  uint32_t *dst = ksk_eng + (size_t)row * rw;
  const int tid = threadIdx.x;
  if (k == 0) {  // unused slots (key.rs:109-111)
    for (int x = tid; x < rw; x += 256) dst[x] = 0u;
    return;
  }
  uint32_t inner = 0;
  for (int x16 = tid; x16 * 16 < n; x16 += 256) {  // one keystream block = 16 mask words
    uint32_t w[16];
    chacha20_block(key, (uint32_t)x16, row, 0u, 0x4B534Bu /* "KSK" */, w);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int x = x16 * 16 + c;
      if (x < n) {
        dst[x] = w[c];
        inner += key_lv0[x] * w[c];
      }
    }
  }
  // block reduction of the wrapping inner product
  for (int off = 32; off > 0; off >>= 1) inner += __shfl_down(inner, off);
  if ((tid & 63) == 0) s_part[tid >> 6] = inner;
  __syncthreads();
  if (tid == 0) {
    const uint32_t total = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    uint32_t w[16];
    chacha20_block(key, 0u, row, 1u, 0x4B534Bu, w);
    double g0, g1;
    gauss2(w, alpha, g0, g1);
    const double p = (double)((uint32_t)k * key_lv1[i]) / (double)(1u << ((j + 1) * basebit));  // key.rs:113-114
    dst[n] = total + gaussian_torus(p, g0);
    for (int x = n + 1; x < rw; ++x) dst[x] = 0u;
  }
}

// ---- bootstrapping key: key.rs:124-156 ------------------------------------------------
// One wave per TRLWE row (i, r) of TRGSW(s0[i]) (trgsw.rs:29-49):
//   a uniform, b = gaussian(0) + a (*) s1        (trlwe.rs:30-52, product via the FFT as poly_mul does)
//   r <  l: a[0] += s0[i] * f64_to_torus(Bg^-(r+1));  r >= l: b[0] += ... (trgsw.rs:44-47)
//   spectrum of a and b (TRGSWLv1FFT::new, trgsw.rs:58-68) written in engine order with the engine's key scale (key_scale, fft512.hpp).
// s1_spec: forward spectrum of the level-1 key in the forward-FFT bin order (k_key_spectrum).
__global__ __launch_bounds__(64) void k_key_spectrum(const uint32_t *__restrict__ key_lv1, const double2 *__restrict__ twt,
                                                      double2 *__restrict__ s1_spec) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *tile = reinterpret_cast<double2 *>(smem);
  const int lane = threadIdx.x;
  Twiddles tw;
  tw.load(twt, reinterpret_cast<double2 *>(smem + kTileBytes), lane);
  double re[8], im[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    re[m] = (double)(int32_t)key_lv1[lane + 64 * m];
    im[m] = (double)(int32_t)key_lv1[lane + 64 * m + kN2];
  }
  fft_forward(re, im, tw, tile, lane);
#pragma unroll
  for (int s = 0; s < 8; ++s) s1_spec[s * 64 + lane] = make_double2(re[s], im[s]);
}

template <int L>
__global__ __launch_bounds__(64) void k_gen_bsk(const uint32_t *__restrict__ key_lv0, const double2 *__restrict__ s1_spec,
                                                 const double2 *__restrict__ twt, double2 *__restrict__ bsk_eng,
                                                 int bgbit, double alpha, const ChaChaKey *__restrict__ key_p,
                                                 double scale /* key_scale(fast) */) {
  const ChaChaKey key = *key_p;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *tile = reinterpret_cast<double2 *>(smem);
  const int lane = threadIdx.x;
  const uint32_t row = blockIdx.x;  // i * 2L + r
  const int r = row % (2 * L), i = row / (2 * L);
  Twiddles tw;
  tw.load(twt, reinterpret_cast<double2 *>(smem + kTileBytes), lane);

  // a: uniform; e: gaussian.  Lane l owns coefficients l+64m (lo) and l+64m+512 (hi), m < 8:
  // 16 mask words per lane = one keystream block, 8 Gaussian pairs = 32 words = two more.
  uint32_t a_lo[8], a_hi[8], b_lo[8], b_hi[8];
  {
    uint32_t w[16];
    chacha20_block(key, (uint32_t)lane, row, 2u, 0x42534Bu /* "BSK" */, w);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a_lo[q] = w[q];
      a_hi[q] = w[8 + q];
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t w[16];
    chacha20_block(key, (uint32_t)(lane * 2 + h), row, 3u, 0x42534Bu, w);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double g0, g1;
      gauss2(w + 4 * q, alpha, g0, g1);
      b_lo[4 * h + q] = gaussian_torus(0.0, g0);
      b_hi[4 * h + q] = gaussian_torus(0.0, g1);
    }
  }
  // poly_res = a (*) s1 (klemsa.rs:152-174): A*S/512 through the inverse
  double re[8], im[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    re[m] = (double)(int32_t)a_lo[m];
    im[m] = (double)(int32_t)a_hi[m];
  }
  fft_forward(re, im, tw, tile, lane);
  double are[8], aim[8];  // keep A for the key spectrum of `a` (linear: gadget added below)
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    are[s] = re[s];
    aim[s] = im[s];
    const double2 sp = s1_spec[s * 64 + lane];
    const double pr = (re[s] * sp.x - im[s] * sp.y) * 0x1p-9;
    const double pi = (re[s] * sp.y + im[s] * sp.x) * 0x1p-9;
    re[s] = pr;
    im[s] = pi;
  }
  fft_inverse(re, im, tw, tile, lane);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    b_lo[m] += round_to_torus<false>(re[m]);
    b_hi[m] += round_to_torus<false>(im[m]);
  }
  // gadget: p * f64_to_torus(Bg^-(d+1)) on coefficient 0 of a (rows < L) or b (rows >= L)
  const uint32_t p = key_lv0[i];
  const int d = r % L;
  const uint32_t gadget = p * dev_f64_to_torus(exp2(-(double)(bgbit * (d + 1))));
  // spectrum of a: FFT is linear and the gadget sits on coefficient 0 = lane 0, slot 0, real part;
  // redo the forward transform only when it changed (r < L) -- cheaper to just transform again.
  if (r < L) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      re[m] = (double)(int32_t)(a_lo[m] + ((lane == 0 && m == 0) ? gadget : 0u));
      im[m] = (double)(int32_t)a_hi[m];
    }
    fft_forward(re, im, tw, tile, lane);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      are[s] = re[s];
      aim[s] = im[s];
    }
  } else if (lane == 0) {
    b_lo[0] += gadget;
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    re[m] = (double)(int32_t)b_lo[m];
    im[m] = (double)(int32_t)b_hi[m];
  }
  fft_forward(re, im, tw, tile, lane);
  double2 *dst = bsk_eng + (size_t)row * 2 * kN2;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    // reference stores 2*DFT (klemsa.rs:112-113); engine folds its key scale (2^-10 or 2^-42, fft512.hpp) on top
    const double k2 = 2.0 * scale;
    dst[s * 64 + lane] = make_double2(are[s] * k2, aim[s] * k2);
    dst[kN2 + s * 64 + lane] = make_double2(re[s] * k2, im[s] * k2);
  }
}

// ---- export: engine layouts back to the reference layouts (tests, key persistence) -------
__global__ void k_bsk_export(const double2 *__restrict__ eng, double *__restrict__ ref, size_t polys, double unscale /* 1 / key_scale(fast) */) {
  size_t p = blockIdx.x;
  int t = threadIdx.x;  // engine position s*64 + mu
  int s = t >> 6, mu = t & 63;
  int k = bin_of(mu, s);
  double2 v = eng[p * kN2 + t];
  ref[p * kN + k] = v.x * unscale;
  ref[p * kN + k + kN2] = v.y * unscale;
}

__global__ void k_ksk_export(const uint32_t *__restrict__ eng, uint32_t *__restrict__ ref, int n, size_t rows) {
  const size_t r = blockIdx.x;
  if (r >= rows) return;
  const int rw = ksk_row_words(n);
  for (int x = threadIdx.x; x <= n; x += blockDim.x) ref[r * (size_t)(n + 1) + x] = eng[r * (size_t)rw + x];
}

}  // namespace tfhe
