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

// ---- base = 4 variant: candidate rows streamed through an LDS ring by async DMA ------------
// With basebit = 2 (the 80/110/128-bit and UINT1 sets) a group (i, j) has only three non-zero
// candidate rows, and all G ciphertexts of a workgroup pick among them.  k_key_switch sends every
// pick through the vector L1 (64 B/clk/CU -- its measured bound).  Here the three rows of a group
// are copied ONCE into LDS by `global_load_lds_dwordx4` (no VGPRs, contiguous 1 KiB per wave
// instruction), NS-1 groups ahead of their use, and every pick is a ds_read_b128 (256 B/clk/CU).
// Hand-off per group: each wave waits for its own DMAs of that group with a COUNTED s_waitcnt
// vmcnt (younger groups stay in flight across the barrier), one s_barrier makes all waves'
// pieces visible and frees the slot of the previous group, which the next DMA then refills.
// The DMA is inline asm so that hipcc neither drains it with vmcnt(0) at LDS reads nor at barriers.
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

#ifndef TFHE_KS_NS
#define TFHE_KS_NS 3
#endif
constexpr int kKsRingSlots = TFHE_KS_NS;  // NS: groups resident in LDS
constexpr int kKsChunksPerWave = 3;  // CW: DMA instructions per wave per group (1 KiB each)

__host__ __device__ __forceinline__ uint32_t ks_b4_slot_bytes(int n) {
  return ((uint32_t)ksk_row_words(n) * 4u * 3u + 1023u) & ~1023u;
}
__host__ __device__ __forceinline__ size_t ks_b4_lds_bytes(int n, int G) {
  // zero row | dummy chunk | ring | a_bar staging
  return (size_t)ksk_row_words(n) * 4 + 1024 + (size_t)kKsRingSlots * ks_b4_slot_bytes(n) + (size_t)G * 64 * 4;
}

template <int G>
__global__ __launch_bounds__(320) void k_key_switch_b4(const uint32_t *__restrict__ lv1,  // [count][N+1]
                                                        const unsigned char *__restrict__ ksk,  // engine layout
                                                        int n, int t, uint32_t *__restrict__ out,
                                                        size_t count) {
  constexpr int N = 1024, IC = 64, NS = kKsRingSlots, D = NS - 1, CW = kKsChunksPerWave;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
  const int rw4 = ksk_row_words(n) >> 2;
  const uint32_t row_bytes = (uint32_t)rw4 * 16u;
  const uint32_t slot_bytes = ks_b4_slot_bytes(n);
  const uint32_t chunks = slot_bytes >> 10;
  // LDS carve (byte offsets from the dynamic base)
  const uint32_t off_zero = 0u, off_dummy = row_bytes, off_ring = row_bytes + 1024u;
  const uint32_t off_ab = off_ring + NS * slot_bytes;
  uint32_t(*s_ab)[IC] = reinterpret_cast<uint32_t(*)[IC]>(ks_smem + off_ab);
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)ks_smem;

  const size_t g0 = (size_t)blockIdx.x * G;
  const int tid = threadIdx.x;
  const int bd = blockDim.x;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t nw = (uint32_t)(bd >> 6);
  const uint32_t lane = (uint32_t)(tid & 63);
  const uint32_t xoff = (tid < rw4 ? (uint32_t)tid : 0u) * 16u;
  const uint32_t prec_offset = 1u << (32 - (1 + 2 * t));
  const uint32_t total = (uint32_t)N * (uint32_t)t;

  for (int x = tid; x < rw4; x += bd) reinterpret_cast<u32x4 *>(ks_smem + off_zero)[x] = u32x4{0u, 0u, 0u, 0u};

  // the three non-zero rows of group q are contiguous: bytes [(4q+1)*row_bytes, (4q+4)*row_bytes)
  auto dma_group = [&](uint32_t q) {
    const uint32_t slot = q % NS;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const uint32_t chunk = wave + (uint32_t)c * nw;  // wave-uniform
      const bool real = (chunk < chunks) & (q < total);
      const size_t gofs = real ? ((size_t)(4u * q + 1u) * row_bytes + (size_t)chunk * 1024u) : 0;
      const uint32_t dst = real ? (off_ring + slot * slot_bytes + chunk * 1024u) : off_dummy;
      glds16(ksk + gofs + lane * 16u, lds_base + dst);  // every wave issues exactly CW DMAs per group
    }
  };

  u32x4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = u32x4{0u, 0u, 0u, 0u};

  uint32_t q = 0;
#pragma unroll 1
  for (uint32_t d = 0; d < (uint32_t)D; ++d) dma_group(d);  // prime: groups 0 .. D-1

#pragma unroll 1
  for (int i0 = 0; i0 < N; i0 += IC) {
    // restage the a_bar words of the next 64 coefficients; drain first so the compiler's own
    // loads below see an empty queue (once per 64*t groups)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int idx = tid; idx < G * IC; idx += bd) {
      const int g = idx / IC, ii = idx % IC;
      const size_t ct = g0 + g;
      s_ab[g][ii] = ct < count ? lv1[ct * (N + 1) + i0 + ii] + prec_offset : 0u;
    }
    __syncthreads();
#pragma unroll 1
    for (int ii = 0; ii < IC; ++ii) {
      uint32_t ab[G];  // wave-uniform: broadcast LDS read -> SGPR
#pragma unroll
      for (int g = 0; g < G; ++g) ab[g] = __builtin_amdgcn_readfirstlane(s_ab[g][ii]);
#pragma unroll 1
      for (int j = 0; j < t; ++j, ++q) {
        // this wave's pieces of group q have landed: at most the D-1 younger groups stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * CW) : "memory");
        __builtin_amdgcn_s_barrier();  // all pieces of q visible; everyone is done with group q-1
        asm volatile("" ::: "memory");
        dma_group(q + (uint32_t)D);    // refill the slot group q-1 just vacated
        const int sh = 32 - (j + 1) * 2;
        const uint32_t slot_off = off_ring + (q % NS) * slot_bytes;
        constexpr int GB = G < 16 ? G : 16;  // LDS reads in flight per lane before their subtractions
#pragma unroll
        for (int gb = 0; gb < G; gb += GB) {
          u32x4 v[GB];
#pragma unroll
          for (int g = 0; g < GB; ++g) {
            const uint32_t k = (ab[gb + g] >> sh) & 3u;
            // roff = k ? slot_off + (k-1)*row_bytes : off_zero (= 0), wave-uniform.  Written as SALU
            // asm: left to itself hipcc does this select per lane (v_mul_lo_u32 + v_cndmask + 2 adds
            // for every pick), which made the kernel VALU-bound.
            uint32_t roff;
            asm("s_sub_u32 %0, %1, 1\n\t"
                "s_mul_i32 %0, %0, %2\n\t"
                "s_add_u32 %0, %0, %3\n\t"
                "s_cmp_eq_u32 %1, 0\n\t"
                "s_cselect_b32 %0, 0, %0"
                : "=&s"(roff)
                : "s"(k), "s"(row_bytes), "s"(slot_off)
                : "scc");
            v[g] = *reinterpret_cast<const u32x4 *>(ks_smem + roff + xoff);
          }
#pragma unroll
          for (int g = 0; g < GB; ++g) acc[gb + g] -= v[g];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (dummy) DMAs before the LDS goes away
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

// ---- small batches: the walk over i split across workgroups ---------------------------------
// The group kernels above walk all N*t (i, j) pairs inside one workgroup -- right for a full
// machine, but ~10 ms of serial latency when only a handful of ciphertexts exist.  Here a
// ciphertext's N coefficients are cut into `gridDim.y` slices; each workgroup subtracts the rows of
// its slice into registers and merges them into the (pre-zeroed) output with integer atomics.
// u32 addition is associative and commutative, so the result is the same bits in any arrival order.
__global__ __launch_bounds__(320) void k_key_switch_split(const uint32_t *__restrict__ lv1,  // [count][N+1]
                                                           const uint4 *__restrict__ ksk,     // engine layout
                                                           uint32_t ksk_bytes, int n, int basebit, int t,
                                                           uint32_t *__restrict__ out) {      // [count][n+1], zeroed
  constexpr int N = 1024;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  const int rw4 = ksk_row_words(n) >> 2;
  const size_t ct = blockIdx.x;
  const int slices = gridDim.y, per = N / slices;  // host picks a divisor of N
  const int i_lo = blockIdx.y * per;
  const int tid = threadIdx.x;
  const uint32_t lane_off = (tid < rw4 ? (uint32_t)tid : 0u) * 16u;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)ksk, 0, (int)ksk_bytes, 0x00020000);
  const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
  const uint32_t base = 1u << basebit, mask = base - 1u;
  const uint32_t row_bytes = (uint32_t)rw4 * 16u;
  u32x4 acc = u32x4{0u, 0u, 0u, 0u};
#pragma unroll 1
  for (int i = i_lo; i < i_lo + per; ++i) {
    const uint32_t ab = (uint32_t)__builtin_amdgcn_readfirstlane(lv1[ct * (N + 1) + i]) + prec_offset;
#pragma unroll 4
    for (int j = 0; j < t; ++j) {
      const uint32_t k = (ab >> (32 - (j + 1) * basebit)) & mask;
      // k == 0 rows are zero in the engine layout: branch-free
      acc -= __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)((((uint32_t)i * t + j) * base + k) * row_bytes), 0);
    }
  }
  if (tid < rw4) {
    uint32_t *o = out + ct * (size_t)(n + 1);
    const uint32_t w[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int x = 4 * tid + c;
      uint32_t v = w[c];
      if (x == n && blockIdx.y == 0) v += lv1[ct * (N + 1) + N];  // res.b = src.b - sum (trgsw.rs:342)
      if (x <= n && v) atomicAdd(o + x, v);
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
