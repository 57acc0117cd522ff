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

#include <type_traits>

#include "experiment.hpp"

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
        const uint32_t bfe_arg = (uint32_t)sh | ((uint32_t)basebit << 16);  // offset | width << 16
        const uint32_t grp_bytes = grp * row_bytes;
        // GB row loads in flight per lane at a time (register budget: acc 4G + v 4GB)
        constexpr int GB = G < 16 ? G : 16;
#pragma unroll
        for (int gb = 0; gb < G; gb += GB) {
          u32x4 v[GB];
#pragma unroll
          for (int g = 0; g < GB; ++g) {
            uint32_t k;  // scalar bit-field extract (hipcc emits shift + and)
            asm("s_bfe_u32 %0, %1, %2" : "=s"(k) : "s"(ab[gb + g]), "s"(bfe_arg) : "scc");
            // one descriptor, lane offset in a VGPR, row offset in an SGPR: no per-lane address math
            v[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)(grp_bytes + k * row_bytes), 0);
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
// are copied ONCE into LDS by `global_load_lds_dwordx4` (no VGPRs, 1 KiB per wave instruction),
// NS-1 groups ahead of their use, and every pick is a ds_read_b128.
// A lane only ever needs its own four columns, so each WAVE copies and consumes its own 1 KiB
// column band of every row: the ring is wave-private, the hand-off is a COUNTED s_waitcnt vmcnt on
// the wave's own DMAs (younger groups stay in flight) and there is no barrier in the group loop.
// What bounds the loop is instruction issue (one per wave per 4 cycles, one scalar unit per CU):
// the pick is kept to s_bfe + s_mul (row = slot + k * stride, each slot carrying its own zero row
// for k = 0) + two vector address adds + ds_read_b128 + 4 v_sub.  r1d spent 261 scalar instructions
// per group on a select-based pick and ran 38 ms; this form issues ~100 and runs 23 ms.
// The DMA is inline asm so that hipcc does not drain it with vmcnt(0) at every LDS read.
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

constexpr int kKsRingSlots = 3;  // groups resident in LDS (measured: 2 / 4 / 5 slots 56 / 40 / 46 ms vs 23 ms)
constexpr int kKsStage = 64;     // coefficients whose a_bar words are staged in LDS at a time (32 / 16: 24.3 / 24.2 vs 24.1 ms)
constexpr uint32_t kKsWaveBytes = 1024;   // one DMA instruction: 64 lanes x 16 B = this wave's columns of one row

// LDS per workgroup of `nw` waves: ring[NS][zero row, 3 rows][nw KiB] | a_bar staging.  Each wave
// owns the 1 KiB column band [w KiB, (w+1) KiB) of every row, so rows sit at a stride of nw KiB; every
// slot starts with its own zero row so that the row picked by digit k is simply slot + k * stride.
__host__ __device__ __forceinline__ uint32_t ks_b4_row_stride(int nw) { return (uint32_t)nw * kKsWaveBytes; }
__host__ __device__ __forceinline__ uint32_t ks_b4_slot_bytes(int nw) { return 4u * ks_b4_row_stride(nw); }
__host__ __device__ __forceinline__ size_t ks_b4_lds_bytes(int nw, int G) {
  return (size_t)kKsRingSlots * ks_b4_slot_bytes(nw) + (size_t)G * kKsStage * 4;
}

template <int G>
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(3))) void k_key_switch_b4(const uint32_t *__restrict__ lv1,  // [count][N+1]
                                                        const unsigned char *__restrict__ ksk,  // engine layout
                                                        int n, int t, uint32_t *__restrict__ out,
                                                        size_t count) {
  constexpr int N = 1024, IC = kKsStage, NS = kKsRingSlots, D = NS - 1;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
  const int rw4 = ksk_row_words(n) >> 2;
  const uint32_t row_bytes = (uint32_t)rw4 * 16u;
  const int tid = threadIdx.x;
  const int bd = blockDim.x;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t nw = (uint32_t)(bd >> 6);
  const uint32_t lane = (uint32_t)(tid & 63);
  const uint32_t row_stride = ks_b4_row_stride((int)nw);
  const uint32_t slot_bytes = 4u * row_stride;
  // LDS carve (byte offsets from the dynamic base)
  const uint32_t off_ring = 0u;
  const uint32_t off_ab = off_ring + NS * slot_bytes;
  uint32_t(*s_ab)[IC] = reinterpret_cast<uint32_t(*)[IC]>(ks_smem + off_ab);
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)ks_smem;

  const size_t g0 = (size_t)blockIdx.x * G;
  const uint32_t xoff = (uint32_t)tid * 16u;  // this lane's 4 columns, in a row and in every LDS copy of one
  const uint32_t prec_offset = 1u << (32 - (1 + 2 * t));
  const uint32_t total = (uint32_t)N * (uint32_t)t;

  for (int sl = 0; sl < NS; ++sl)  // row 0 of every slot: the k == 0 pick
    for (int x = tid; x < (int)(row_stride >> 4); x += bd)
      reinterpret_cast<u32x4 *>(ks_smem + off_ring + sl * slot_bytes)[x] = u32x4{0u, 0u, 0u, 0u};

  // Group q = (i, j): its three non-zero rows are rows 4q+1 .. 4q+3 of the key.  This wave copies ITS
  // column band of each (bytes past the row end belong to the next row or the allocation's tail pad and
  // only ever reach lanes whose columns are never stored).  Past the last group the source is clamped:
  // every wave still issues exactly three DMAs per group, so the counted waits below stay exact.
  auto dma_group = [&](uint32_t q) {
    const uint32_t slot = q % NS;
    const uint32_t qs = q < total ? q : total - 1u;
#pragma unroll
    for (uint32_t c = 0; c < 3u; ++c) {
      const size_t gofs = (size_t)(4u * qs + 1u + c) * row_bytes + (size_t)wave * kKsWaveBytes;
      const uint32_t dst = off_ring + slot * slot_bytes + (c + 1u) * row_stride + wave * kKsWaveBytes;
      glds16(ksk + gofs + lane * 16u, lds_base + dst);
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
        // this wave's band of group q has landed (at most the D-1 younger groups stay in flight); no
        // other wave reads or writes it, so there is no barrier; the slot refilled next is the one this
        // wave finished reading in the previous iteration (its subtractions consumed the reads)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 3) : "memory");
        dma_group(q + (uint32_t)D);
        const int sh = 32 - (j + 1) * 2;
        const uint32_t slot_off = off_ring + (q % NS) * slot_bytes;
        const uint32_t bfe_arg = (uint32_t)sh | (2u << 16);  // offset | width << 16
        constexpr int GB = G < 8 ? G : 8;  // LDS reads in flight per lane before their subtractions (registers: 3 waves per SIMD)
#pragma unroll
        for (int gb = 0; gb < G; gb += GB) {
          u32x4 v[GB];
#pragma unroll
          for (int g = 0; g < GB; ++g) {
            // wave-uniform pick: two scalar instructions (bfe, mul) and two vector address adds per row
            uint32_t k;  // s_bfe_u32 (hipcc emits shift + and, or v_bfe_u32 for the builtin)
            asm("s_bfe_u32 %0, %1, %2" : "=s"(k) : "s"(ab[gb + g]), "s"(bfe_arg) : "scc");
            const uint32_t roff = slot_off + k * row_stride;
            v[g] = *reinterpret_cast<const u32x4 *>(ks_smem + roff + xoff);
          }
#pragma unroll
          for (int g = 0; g < GB; ++g) acc[gb + g] -= v[g];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (clamped) DMAs before the LDS goes away
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

// ---- column-sliced kernel for wider bases (basebit 4..7: the UINT2..UINT8 sets) ---------------------------------
// With base = 32 nearly every ciphertext of a group picks a different candidate row, so k_key_switch fetches a whole
// 3.3 KB row from L2 for every (ciphertext, group): 662 GB per 65,536-batch at SECURITY_UINT4, L2-bandwidth bound
// (49 ms).  Here a workgroup owns a SLICE of 64 output columns for 16 * S ciphertexts: all `base` candidate rows of a
// group, cut to that slice (base x 256 B), are copied once into an LDS ring by global_load_lds and every ciphertext
// picks from LDS.  L2 traffic falls by the ciphertexts-per-workgroup ratio (42 GB at UINT4).
// Lane map: 16 lanes cover the 64 columns (4 each); the 4 lane-quarters of a wave serve 4 different ciphertexts per
// instruction (ds_read_b128 is serviced in 16-lane groups, so different rows per quarter cost nothing); S accumulator
// sets per lane -> 4*S ciphertexts per wave, 16*S per workgroup of 4 waves.
// Waves per workgroup: four, two workgroups per CU, while two rings fit a CU's LDS (base 16 / 32); EIGHT in ONE workgroup
// per CU at base 64 / 128, whose rings (16 / 32 KiB per group) leave room for one workgroup only -- the same two waves
// per SIMD, one ring and one digit stage for twice the ciphertexts (half the L2 -> LDS traffic per ciphertext).
#ifndef TFHE_KS_SL2_WAVES_WIDE  // (experiment knob: 4 = the four-wave workgroup at base 64 / 128 too)
#define TFHE_KS_SL2_WAVES_WIDE 8
#endif
#ifndef TFHE_KS_SL2_WAVES_B32  // (experiment knob: 8 = one eight-wave workgroup per CU at base 32 as well; base 16's 4 KiB groups are 4 DMAs)
#define TFHE_KS_SL2_WAVES_B32 4
#endif
__host__ __device__ constexpr int ks_sl2_waves(int basebit) { return basebit <= 4 ? 4 : basebit == 5 ? TFHE_KS_SL2_WAVES_B32 : TFHE_KS_SL2_WAVES_WIDE; }
__host__ __device__ constexpr int ks_sl2_cts(int basebit, int sets) { return 4 * sets * ks_sl2_waves(basebit); }  // ciphertexts per workgroup (512 at S = 32, four waves)

// The kernel's first form (rounds 1-3: digits extracted per read with v_bfe + v_lshl_add, four v_sub per row read, a_bar
// words restaged every 16 coefficients behind a drained DMA queue, one barrier per group; 12.4-13.5 ms for 65,536
// ciphertexts at SECURITY_UINT4) was VALU-issue bound -- profiles/exp/logs/r4_ks_sl_ablation.log: with the key DMA,
// the barrier, the restage and the LDS reads all removed it still took 7.5 ms.  What follows replaced it (8.2 ms).
// Pre-digested digits, v_perm picks, v_add3 over pairs of groups:
//  * the digits are extracted ONCE per (ciphertext, group) by a streaming pre-pass (k_ks_digits) into one byte each,
//    four groups to a word, laid out [quad of groups][ciphertext]: the kernel DMAs a quad's words for its ciphertexts
//    straight into LDS (global_load_lds_dword, no registers, no restage stall) one quad ahead;
//  * the pick is ONE v_perm_b32: LDS address = { 0, 0, digit byte j of the word, lane's column byte } -- the row
//    stride is 256 B, so the digit byte IS address bits 8..15 -- and the ring slot is the ds_read's immediate offset
//    (the loop is unrolled over the ring, so slot numbers are compile-time);
//  * groups are consumed in PAIRS: acc = v_add3_u32(acc, row_g, row_g+1), two VALU instructions per row read where
//    the first form spends four (the sum is negated once at the end: res = src.b - sum, trgsw.rs:342-356);
//  * one barrier per pair of groups instead of one per group; the ring holds RP pairs (lookahead RP-1 pairs).
// VALU per row read: 1 (perm) + 2 (add3) = 3 instead of 6.  Same lane map, same ring layout, same K chunks.
template <int BASEBIT>
struct KsSl2 {
  static constexpr int BASE = 1 << BASEBIT;
  static constexpr int SLOT = BASE * 256;            // bytes of one group's slice: BASE rows x 64 columns
  static constexpr int CW = ks_sl2_waves(BASEBIT);   // waves per workgroup
  static constexpr int PWG = SLOT / 1024 / CW;       // ring DMA instructions per wave per group (1 KiB each)
  static_assert(PWG >= 1, "a group's slice must be at least one DMA instruction per wave (base >= 16)");
};
// stage words per quad: one per ciphertext, padded so that every wave issues the same whole number of 256-byte DMAs
__host__ __device__ constexpr int ks_sl2_stage_words(int basebit, int sets) {
  return (ks_sl2_cts(basebit, sets) + 64 * ks_sl2_waves(basebit) - 1) / (64 * ks_sl2_waves(basebit)) * (64 * ks_sl2_waves(basebit));
}
__host__ __device__ constexpr size_t ks_sl2_lds_bytes(int basebit, int sets, int rp) {
  return (size_t)2 * rp * ((size_t)(1 << basebit) * 256) + (size_t)2 * ks_sl2_stage_words(basebit, sets) * 4;
}
// pairs of groups in the ring: 4 (lookahead 6 groups) at base 16 / 32, 2 at base 64 / 128
#ifndef TFHE_KS_SL2_RP_SMALL  // (experiment knob)
#define TFHE_KS_SL2_RP_SMALL 4
#endif
#ifndef TFHE_KS_SL2_RP_B64  // (experiment knob) base 64: 2 pairs (64 KiB); 4 pairs measured no better (11.1 vs 10.9 ms at SECURITY_UINT6)
#define TFHE_KS_SL2_RP_B64 2
#endif
__host__ __device__ constexpr int ks_sl2_rp(int basebit) { return basebit <= 5 ? TFHE_KS_SL2_RP_SMALL : basebit == 6 ? TFHE_KS_SL2_RP_B64 : 2; }
// rows of the digit buffer: the ciphertexts of whole workgroups plus one DMA round of slack
__host__ __device__ inline size_t ks_sl2_ct_stride(size_t count, int basebit, int sets) {
  const size_t cts = (size_t)ks_sl2_cts(basebit, sets);
  return (count + cts - 1) / cts * cts + 512;
}

// The same DMAs with the address split as (wave-uniform 64-bit base in SGPRs) + (per-lane 32-bit offset in ONE VGPR that
// never changes): no per-instruction vector address arithmetic, no address registers held across the loop.
__device__ __forceinline__ void glds16_s(const void *sbase, uint32_t voff, uint32_t lds_dst) {  // 64 lanes x 16 B -> LDS[m0 + 16 * lane]
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ void glds4_s(const void *sbase, uint32_t voff, uint32_t lds_dst) {  // 64 lanes x 4 B -> LDS[m0 + 4 * lane]
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// digit pre-pass: lv1 [count][N+1] -> digw [N*t/4][ct_stride] u32, byte b of word (Q, ct) = digit of group 4Q + b
// (group g = coefficient g / t, level g % t; digit = ((a + prec_offset) >> (32 - (j+1)*basebit)) & (base-1),
// trgsw.rs:343-349); rows past `count` are all-zero digits (the k = 0 row of every group is zero).
// One block = 64 ciphertexts x 32 quads: the level-1 words are read along the rows (coalesced), turned in LDS, and
// the digit words leave ciphertext-major (coalesced).
__global__ __launch_bounds__(256) void k_ks_digits(const uint32_t *__restrict__ lv1, uint32_t *__restrict__ digw,
                                                    size_t ct_stride, size_t count, int basebit, int t) {
  constexpr int N = 1024, QB = 32, CB = 64, WMAX = QB * 4 + 2;  // words of a row a block may need: 128 groups / t (+ ends)
  __shared__ uint32_t tile[CB][WMAX + 1];
  const size_t ct0 = (size_t)blockIdx.x * CB;
  const int q0 = blockIdx.y * QB;           // first quad of this block
  const int g0 = q0 * 4, g1 = g0 + QB * 4;  // groups [g0, g1)
  const int i_lo = g0 / t, i_hi = (g1 - 1) / t;  // coefficients needed
  const int nw = i_hi - i_lo + 1;
  const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
  for (int idx = threadIdx.x; idx < CB * nw; idx += 256) {
    const int r = idx / nw, w = idx % nw;
    const size_t ct = ct0 + r;
    tile[r][w] = ct < count ? lv1[ct * (N + 1) + i_lo + w] + prec_offset : 0u;
  }
  __syncthreads();
  const uint32_t mask = (1u << basebit) - 1u;
  for (int idx = threadIdx.x; idx < CB * QB; idx += 256) {
    const int r = idx % CB, q = idx / CB;
    const size_t ct = ct0 + r;
    uint32_t word = 0;
    if (ct < count) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int g = g0 + q * 4 + b, i = g / t, j = g - i * t;
        const uint32_t k = (tile[r][i - i_lo] >> (32 - (j + 1) * basebit)) & mask;
        word |= k << (8 * b);
      }
    }
    if (ct < ct_stride) digw[(size_t)(q0 + q) * ct_stride + ct] = word;
  }
}

__device__ __forceinline__ void add3_inplace(uint32_t &acc, uint32_t a, uint32_t b) {
  asm("v_add3_u32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int BASEBIT, int S, int RP>
__global__ __launch_bounds__(64 * ks_sl2_waves(BASEBIT), 2) void k_key_switch_sliced(const uint32_t *__restrict__ digw, size_t ct_stride,
                                                             const uint32_t *__restrict__ lv1,       // [count][N+1]: the body word
                                                             const unsigned char *__restrict__ ksk,  // engine layout
                                                             int n, int t, uint32_t *__restrict__ out, size_t count) {
  using G = KsSl2<BASEBIT>;
  constexpr int N = 1024, BASE = G::BASE, SLOT = G::SLOT, PWG = G::PWG, CW = G::CW, RG = 2 * RP, CTS = ks_sl2_cts(BASEBIT, S);
  constexpr int STW = ks_sl2_stage_words(BASEBIT, S);  // stage words per quad (one per ciphertext, padded)
  constexpr int ST = STW / 64 / CW;                    // stage DMA instructions per wave per quad (256 B each)
  constexpr int PW = 2 * PWG;                      // ring DMA instructions per wave per pair
  static_assert(RP == 2 || RP == 4, "ring of 2 or 4 pairs");
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  extern __shared__ __attribute__((aligned(16))) unsigned char sl_smem[];
  constexpr uint32_t off_stage = (uint32_t)RG * SLOT;
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)sl_smem;
  if (lds_base != 0u) __builtin_trap();  // (the row reads below address the ring from 0)
  const uint32_t row_bytes = (uint32_t)ksk_row_words(n) * 4u;

  const int tid = threadIdx.x;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane = (uint32_t)(tid & 63);
  const uint32_t sub = lane >> 4, c4 = lane & 15u;
  const size_t ct0 = (size_t)blockIdx.x * CTS;
  const uint32_t col0 = blockIdx.y * 64u;
  const uint32_t total = (uint32_t)N * (uint32_t)t;  // groups
  const uint32_t kchunks = gridDim.z, gpz = total / kchunks;  // groups of this workgroup: a multiple of RG (host)
  const uint32_t g_begin = blockIdx.z * gpz, g_end = g_begin + gpz;

  // group g -> ring slot g % RG; instruction x of a group moves rows 4x .. 4x+3 (lane quarter = row).  Past the range
  // the source is clamped (the bytes land in slots nobody reads again); every wave issues exactly PWG per group.
  const uint32_t voff_ring = sub * row_bytes + c4 * 16u;  // this lane's row of the four an instruction moves, its 16 bytes
  const uint32_t voff_stage = lane * 4u;
  auto dma_group = [&](uint32_t g) {
    const uint32_t slot = g % RG;
    const uint32_t gs = g < total ? g : total - 1u;  // (rows gs * BASE + 4x + sub < total * BASE: always inside the key)
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)PWG; ++c) {
      const uint32_t x = wave + c * CW;
      glds16_s(ksk + (size_t)(gs * BASE + 4u * x) * row_bytes + col0 * 4u, voff_ring, lds_base + slot * SLOT + x * 1024u);
    }
  };
  // quad Q -> stage buffer Q & 1: one word per ciphertext of this workgroup (padded to STW)
  auto dma_stage = [&](uint32_t Q) {
    const uint32_t Qs = Q < total / 4u ? Q : total / 4u - 1u;
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)ST; ++c) {
      const uint32_t y = wave + c * CW;
      glds4_s(digw + (size_t)Qs * ct_stride + ct0 + y * 64u, voff_stage, lds_base + off_stage + (Q & 1u) * (STW * 4u) + y * 256u);
    }
  };

  uint32_t acc[S][4];
#pragma unroll
  for (int a = 0; a < S; ++a) acc[a][0] = acc[a][1] = acc[a][2] = acc[a][3] = 0u;
  const uint32_t lane_col = c4 * 16u | 0x10000u;  // address byte 0 (byte 1 comes from the digit); byte 2 = 1 for the slots beyond 64 KiB
  const uint32_t my_ct = wave * (4u * S) + sub;  // + 4a: this lane quarter's ciphertexts within the workgroup

  // prologue, in the steady-state order [stage][ring]: the first quad's digits, ring pairs 0 .. RP-2
  dma_stage(g_begin / 4u);
#pragma unroll
  for (uint32_t p = 0; p + 1 < (uint32_t)RP; ++p) {
    dma_group(g_begin + 2u * p);
    dma_group(g_begin + 2u * p + 1u);
  }

  uint32_t abw[S];
  // one step = one pair of groups; `pc` carries the step's position in the ring as a compile-time constant (slot
  // offsets, digit byte selectors and the counted waits are immediates)
  auto step = [&](uint32_t g, auto pc) {
    constexpr int p = decltype(pc)::value;
    {
      // Step: this wave's pieces of the pair (issued RP-1 steps ago) have landed when at most what it issued since is
      // outstanding: (RP-2) further pairs and the one stage batch that falls into those steps; on a quad's first
      // step the quad's digits (issued two steps ago, BEFORE that step's pair) must have landed too.
      if (RP == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (p % 2 == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RP - 2) * PW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RP - 2) * PW + ST) : "memory");
      if (!(TFHE_ABL_SL & 2)) __builtin_amdgcn_s_barrier();  // every wave's pieces are visible; the pair consumed last step is retired
      asm volatile("" ::: "memory");
      const uint32_t quad = (g + 2u * p) / 4u;
      if (p % 2 == 0) dma_stage(quad + 1u);
      if (!(TFHE_ABL_SL & 1)) {
        dma_group(g + 2u * (p + RP - 1));
        dma_group(g + 2u * (p + RP - 1) + 1u);
      }
      if (p % 2 == 0) {  // the quad's digit words of this lane quarter's ciphertexts
        const uint32_t *st = reinterpret_cast<const uint32_t *>(sl_smem + off_stage + (quad & 1u) * (STW * 4u));
#pragma unroll
        for (int a = 0; a < S; ++a) abw[a] = st[my_ct + 4u * a];
      }
      // slot offsets ride in the ds_read's 16-bit immediate; a slot at or beyond 64 KiB (base 128: slots of 32 KiB) takes
      // address bit 16 from byte 2 of the lane constant instead (selector byte 2 = 0x02), so it costs no instruction either
      constexpr uint32_t s0 = (uint32_t)((2 * p) % RG) * SLOT, s1 = (uint32_t)((2 * p + 1) % RG) * SLOT;
      constexpr uint32_t slot0 = s0 & 0xFFFFu, slot1 = s1 & 0xFFFFu;
      static_assert(s0 < 0x20000u && s1 < 0x20000u, "ring within 128 KiB");
      constexpr uint32_t sel0 = (s0 >= 0x10000u ? 0x0C020400u : 0x0C0C0400u) + ((uint32_t)((2 * p) % 4) << 8);
      constexpr uint32_t sel1 = (s1 >= 0x10000u ? 0x0C020400u : 0x0C0C0400u) + ((uint32_t)((2 * p + 1) % 4) << 8);
      // Software pipeline over batches of GB ciphertexts: the picks and row reads of batch b+1 are issued BEFORE the
      // additions of batch b, so the LDS always has this wave's next reads queued while its VALU adds (with the reads
      // drained per batch the two pipes took turns: 8.4 ms = VALU 3.8 + LDS 4.6 at SECURITY_UINT4).  The batches are
      // fenced for the scheduler: hoisting still more reads over additions only costs registers.
#ifndef TFHE_KS_SL2_GB
#define TFHE_KS_SL2_GB 2
#endif
      constexpr int GB = TFHE_KS_SL2_GB, NB = S / GB;
      static_assert(S % GB == 0, "the read batch must divide S");
      u32x4 r0[2][GB], r1[2][GB];
      auto issue = [&](int bt, int buf) {
#pragma unroll
        for (int x = 0; x < GB; ++x) {
          const uint32_t a0 = __builtin_amdgcn_perm(abw[bt * GB + x], lane_col, sel0);
          const uint32_t a1 = __builtin_amdgcn_perm(abw[bt * GB + x], lane_col, sel1);
          // LDS addresses as integers: the dynamic LDS of a kernel without static LDS starts at 0 (checked on entry), so
          // the picked address goes to the ds_read as it is and the slot rides in the instruction's offset field
          if (TFHE_ABL_SL & 8) {  // timing-only: no LDS row reads
            r0[buf][x] = u32x4{a0, a1, a0 ^ a1, a0 + 1u};
            r1[buf][x] = u32x4{a1, a0, a1 + 1u, a0 & a1};
            continue;
          }
          r0[buf][x] = *(const __attribute__((address_space(3))) u32x4 *)(uintptr_t)(a0 + slot0);
          r1[buf][x] = *(const __attribute__((address_space(3))) u32x4 *)(uintptr_t)(a1 + slot1);
        }
      };
      issue(0, 0);
#pragma unroll
      for (int bt = 0; bt < NB; ++bt) {
        __builtin_amdgcn_sched_barrier(0);
        if (bt + 1 < NB) issue(bt + 1, (bt + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < GB; ++x) {
          if (TFHE_ABL_SL & 16) {  // timing-only: the rows are read and dropped
            asm volatile("" ::"v"(r0[bt & 1][x]), "v"(r1[bt & 1][x]));
            continue;
          }
          // v_add3_u32 x 4, IN PLACE (inline asm: left to itself the allocator rotates the accumulators through the
          // row buffers' registers across the unrolled steps and needs ~60 more of them; S = 32 / 36 then spill)
          add3_inplace(acc[bt * GB + x][0], r0[bt & 1][x].x, r1[bt & 1][x].x);
          add3_inplace(acc[bt * GB + x][1], r0[bt & 1][x].y, r1[bt & 1][x].y);
          add3_inplace(acc[bt * GB + x][2], r0[bt & 1][x].z, r1[bt & 1][x].z);
          add3_inplace(acc[bt * GB + x][3], r0[bt & 1][x].w, r1[bt & 1][x].w);
        }
      }
    }
  };
#pragma unroll 1
  for (uint32_t g = g_begin; g < g_end; g += RG) {  // one ring's worth of groups: RP steps
    step(g, std::integral_constant<int, 0>{});
    step(g, std::integral_constant<int, 1>{});
    if constexpr (RP == 4) {
      step(g, std::integral_constant<int, 2>{});
      step(g, std::integral_constant<int, 3>{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (clamped) DMAs before the LDS goes away
#pragma unroll
  for (int a = 0; a < S; ++a) {
    const size_t ct = ct0 + my_ct + 4u * a;
    if (ct < count) {
      uint32_t *o = out + ct * (size_t)(n + 1);
      const uint32_t w[4] = {0u - acc[a][0], 0u - acc[a][1], 0u - acc[a][2], 0u - acc[a][3]};  // res = src.b - sum
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int x = (int)(col0 + c4 * 4u) + c;
        if (kchunks == 1) {
          if (x < n) o[x] = w[c];
          if (x == n) o[x] = w[c] + lv1[ct * (N + 1) + N];  // trgsw.rs:342
        } else if (x <= n) {
          const uint32_t v = w[c] + ((x == n && blockIdx.z == 0) ? lv1[ct * (N + 1) + N] : 0u);
          if (v) atomicAdd(&o[x], v);
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

// Proxy re-encryption (proxy_reenc.rs:468-510) is the identity key switch with a source of n coefficients instead of N
// and rides on the kernels above: a level-0 sample [n+1] becomes a level-1 shaped row [N+1] whose coefficients n .. N-1
// are 0 -- a_bar = prec_offset there, every digit 0, no row subtracted (trgsw.rs:343-351) -- with its body word at N.
__global__ __launch_bounds__(256) void k_reenc_pad(const uint32_t *__restrict__ in, uint32_t *__restrict__ lv1, int n) {
  constexpr int N = 1024;
  const uint32_t *src = in + (size_t)blockIdx.x * (size_t)(n + 1);
  uint32_t *dst = lv1 + (size_t)blockIdx.x * (size_t)(N + 1);
  for (int i = threadIdx.x; i <= N; i += 256) dst[i] = i < n ? src[i] : (i == N ? src[n] : 0u);
}

}  // namespace tfhe
