// key_switch.hpp -- identity key switching (src/trgsw.rs:332-360), integer, bit-exact.
//
// res.b = src.b; for i<N: a_bar = a_i + 2^(32-(1+basebit*t)); for j<t:
//   k = (a_bar >> (32-(j+1)*basebit)) & (base-1); if k != 0: res -= KSK[base*t*i + base*j + k]
//
// Engine layout of the key: [N][t][base][RW] u32 with RW = (n+1) rounded up to 4
// words, so every row is 16-byte aligned and read as one dwordx4 per lane; the
// k == 0 rows (never read by the reference, key.rs:107-118) and the pad words are
// zero, which makes the inner loop branch-free: every (i, j, ciphertext) is one
// coalesced row load and one 4-wide subtraction.
//
// Mapping: a workgroup owns G ciphertexts; lane x owns output words 4x..4x+3 of
// all G, accumulated in registers.  The walk over (i, j) is shared by the G
// ciphertexts, so the `base` candidate rows of one (i, j) stream through L1/L2
// once per group instead of once per ciphertext.  The digits are wave-uniform:
// a_bar comes in through scalar loads and the row address is an SGPR base.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfhe {

__host__ __device__ __forceinline__ int ksk_row_words(int n) { return (n + 1 + 3) & ~3; }

template <int G>
__global__ __launch_bounds__(320) void k_key_switch(const uint32_t *__restrict__ lv1,  // [count][N+1]
                                                     const uint4 *__restrict__ ksk,     // engine layout
                                                     uint32_t ksk_bytes, int n, int basebit, int t,
                                                     uint32_t *__restrict__ out,  // [count][n+1]
                                                     size_t count) {
  constexpr int N = 1024;
  constexpr int IC = 64;  // coefficients staged per chunk
  __shared__ uint32_t s_ab[G][IC];
  const int rw4 = ksk_row_words(n) >> 2;
  const size_t g0 = (size_t)blockIdx.x * G;
  const int tid = threadIdx.x;
  const int bd = blockDim.x;
  const uint32_t lane_off = (tid < rw4 ? (uint32_t)tid : 0u) * 16u;  // idle lanes shadow lane 0, never store
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)ksk, 0, (int)ksk_bytes, 0x00020000);
  const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
  const uint32_t base = 1u << basebit;
  const uint32_t mask = base - 1u;

  uint4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = make_uint4(0u, 0u, 0u, 0u);

#pragma unroll 1
  for (int i0 = 0; i0 < N; i0 += IC) {
    __syncthreads();
    for (int idx = tid; idx < G * IC; idx += bd) {
      const int g = idx / IC, ii = idx % IC;
      const size_t ct = g0 + g;
      // past-the-end slots get a_bar = 0: every digit 0 -> the all-zero k = 0 row
      s_ab[g][ii] = ct < count ? lv1[ct * (N + 1) + i0 + ii] + prec_offset : 0u;
    }
    __syncthreads();
#pragma unroll 1
    for (int ii = 0; ii < IC; ++ii) {
      uint32_t ab[G];  // wave-uniform: broadcast LDS read -> SGPR
#pragma unroll
      for (int g = 0; g < G; ++g) ab[g] = __builtin_amdgcn_readfirstlane(s_ab[g][ii]);
      // byte offset of row (i, j, k): (((i*t + j)*base + k) * RW) * 4 < 2^32 for every supported set
      const uint32_t row_bytes = (uint32_t)rw4 * 16u;
      uint32_t grp = (uint32_t)(i0 + ii) * (uint32_t)t * base;  // row index of (i, j=0, k=0)
#pragma unroll 1
      for (int j = 0; j < t; ++j, grp += base) {
        const int sh = 32 - (j + 1) * basebit;
        // GB row loads in flight per lane at a time (register budget: acc 4G + v 4GB)
        constexpr int GB = G < 16 ? G : 16;
#pragma unroll
        for (int gb = 0; gb < G; gb += GB) {
          u32x4 v[GB];
#pragma unroll
          for (int g = 0; g < GB; ++g) {
            const uint32_t k = (ab[gb + g] >> sh) & mask;
            // one descriptor, lane offset in a VGPR, row offset in an SGPR: no per-lane address math
            v[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)((grp + k) * row_bytes), 0);
          }
#pragma unroll
          for (int g = 0; g < GB; ++g) {
            acc[gb + g].x -= v[g].x;
            acc[gb + g].y -= v[g].y;
            acc[gb + g].z -= v[g].z;
            acc[gb + g].w -= v[g].w;
          }
        }
      }
    }
  }
  if (tid < rw4) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const size_t ct = g0 + g;
      if (ct < count) {
        uint32_t *o = out + ct * (size_t)(n + 1);
        const uint32_t w[4] = {acc[g].x, acc[g].y, acc[g].z, acc[g].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int x = 4 * tid + c;
          if (x < n) o[x] = w[c];
          if (x == n) o[x] = w[c] + lv1[ct * (N + 1) + N];  // res.b = src.b - sum (trgsw.rs:342)
        }
      }
    }
  }
}

// reference layout [N*t*base][n+1] -> engine layout [N*t*base][RW], k == 0 rows and pads zeroed
__global__ void k_ksk_convert(const uint32_t *__restrict__ ref, uint32_t *__restrict__ eng, int n, int base,
                              size_t rows) {
  const size_t r = blockIdx.x;
  if (r >= rows) return;
  const int rw = ksk_row_words(n);
  const bool zero = (r % (size_t)base) == 0;
  const uint32_t *src = ref + r * (size_t)(n + 1);
  uint32_t *dst = eng + r * (size_t)rw;
  for (int x = threadIdx.x; x < rw; x += blockDim.x) dst[x] = (zero || x > n) ? 0u : src[x];
}

}  // namespace tfhe
