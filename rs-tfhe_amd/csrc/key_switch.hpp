// key_switch.hpp -- identity key switching (src/trgsw.rs:332-360), integer, bit-exact.
//
// res.b = src.b; for i<N: a_bar = a_i + 2^(32-(1+basebit*t)); for j<t:
//   k = (a_bar >> (32-(j+1)*basebit)) & (base-1); if k != 0: res -= KSK[base*t*i + base*j + k]
//
// Mapping: a workgroup owns G ciphertexts; thread x owns output coordinate(s)
// x (+ blockDim) of all G, accumulating in registers.  The walk over (i, j) is
// shared by the G ciphertexts, so the `base` candidate rows of one (i, j) are
// pulled through L1/L2 once per group instead of once per ciphertext.  The
// k == 0 rows of the uploaded key are zeroed at upload (the reference never
// reads them, key.rs:107-118), which makes the inner loop branch-free: every
// (i, j, g) is one coalesced row load and one subtraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfhe {

template <int G, int XC>
__global__ void k_key_switch(const uint32_t *__restrict__ lv1,  // [count][N+1]
                             const uint32_t *__restrict__ ksk,  // [N][t][base][n+1], k=0 rows zero
                             int n, int basebit, int t, uint32_t *__restrict__ out, size_t count) {
  constexpr int N = 1024;
  extern __shared__ uint32_t s_abar[];  // [G][N]
  const size_t g0 = (size_t)blockIdx.x * G;
  const int tid = threadIdx.x;
  const int bd = blockDim.x;
  const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
  for (int idx = tid; idx < G * N; idx += bd) {
    size_t ct = g0 + (size_t)(idx / N);
    s_abar[idx] = ct < count ? lv1[ct * (N + 1) + (idx % N)] + prec_offset : 0u;
  }
  uint32_t acc[G][XC];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      int x = tid + c * bd;
      size_t ct = g0 + g;
      acc[g][c] = (x == n && ct < count) ? lv1[ct * (N + 1) + N] : 0u;
    }
  __syncthreads();
  const int base = 1 << basebit;
  const uint32_t mask = (uint32_t)base - 1u;
  const size_t row = (size_t)(n + 1);
  int xs[XC];
#pragma unroll
  for (int c = 0; c < XC; ++c) {
    int x = tid + c * bd;
    xs[c] = x <= n ? x : n;  // clamp: out-of-range lanes read a valid word, never stored
  }
  for (int i = 0; i < N; ++i) {
    uint32_t ab[G];
#pragma unroll
    for (int g = 0; g < G; ++g) ab[g] = s_abar[g * N + i];
    const uint32_t *rows_i = ksk + (size_t)i * t * base * row;
    for (int j = 0; j < t; ++j) {
      const int sh = 32 - (j + 1) * basebit;
      const uint32_t *rows = rows_i + (size_t)j * base * row;
      uint32_t v[G][XC];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const uint32_t k = (ab[g] >> sh) & mask;
        const uint32_t *r = rows + (size_t)k * row;
#pragma unroll
        for (int c = 0; c < XC; ++c) v[g][c] = r[xs[c]];
      }
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int c = 0; c < XC; ++c) acc[g][c] -= v[g][c];
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    size_t ct = g0 + g;
    if (ct >= count) continue;
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      int x = tid + c * bd;
      if (x <= n) out[ct * row + x] = acc[g][c];
    }
  }
}

// zero the k == 0 rows of an uploaded key-switching key (key.rs:107-118: unused slots)
__global__ void k_ksk_zero_k0(uint32_t *ksk, int n, int base, size_t groups) {
  size_t grp = blockIdx.x;  // (i, j)
  if (grp >= groups) return;
  uint32_t *r = ksk + grp * (size_t)base * (size_t)(n + 1);
  for (int x = threadIdx.x; x <= n; x += blockDim.x) r[x] = 0u;
}

}  // namespace tfhe
