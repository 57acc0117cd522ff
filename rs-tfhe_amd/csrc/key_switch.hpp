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

// ---- column-sliced variant for wider bases (basebit 3..7: the UINT2..UINT7 sets) ---------------------
// With base = 32 nearly every ciphertext of a group picks a different candidate row, so k_key_switch
// fetches a whole 3.3 KB row from L2 for every (ciphertext, group): 662 GB per 65,536-batch at
// SECURITY_UINT4, L2-bandwidth bound (49 ms).  Here a workgroup owns a SLICE of 64 output columns for
// 512 ciphertexts: all `base` candidate rows of a group, cut to that slice (base x 256 B), are copied
// once into an LDS ring by global_load_lds and every ciphertext picks from LDS.  L2 traffic falls by
// the ciphertexts-per-workgroup ratio (42 GB at UINT4).
// Lane map: 16 lanes cover the 64 columns (4 each); the 4 lane-quarters of a wave serve 4 different
// ciphertexts per instruction (ds_read_b128 is serviced in 16-lane groups, so different rows per quarter
// cost nothing); S accumulator sets per lane -> 4*S ciphertexts per wave, 16*S per workgroup of 4 waves.
// The digit is per quarter, so the pick is two VALU instructions (v_bfe_u32, v_lshl_add_u32).
constexpr int kKsSlSets = 32;      // S: accumulator sets per lane (the default; the host picks S per launch, below)
constexpr int kKsSlWaves = 4;
__host__ __device__ constexpr int ks_sliced_cts(int sets) { return 4 * sets * kKsSlWaves; }  // ciphertexts per workgroup (512 at S = 32)
constexpr int kKsSlCts = ks_sliced_cts(kKsSlSets);
// coefficients whose a_bar words are staged in LDS at a time: 16, or 8 where the ring itself is large
// (base 64: 3 x 16 KiB), so that two workgroups still share a CU's LDS
__host__ __device__ __forceinline__ int ks_sliced_stage(int base) { return base >= 64 ? 8 : 16; }
constexpr int kKsSlSlots = 3;      // ring depth

__host__ __device__ __forceinline__ uint32_t ks_sliced_slot_bytes(int base) {
  return (uint32_t)((base + 15) & ~15) * 256u;  // whole DMA instructions: 4 rows each, 4 waves
}
__host__ __device__ __forceinline__ size_t ks_sliced_lds_bytes(int base, int sets = kKsSlSets) {
  return (size_t)kKsSlSlots * ks_sliced_slot_bytes(base) + (size_t)ks_sliced_cts(sets) * ks_sliced_stage(base) * 4;
}
// S is chosen per launch so that the grid fills whole rounds of the machine: a workgroup's time is proportional to
// S, the grid is ceil(count / 16S) x slices workgroups, `slots` of them run at once, so the launch costs
// ceil(grid / slots) x S.  At SECURITY_UINT4 (65,536 ciphertexts, 13 slices, 512 slots) S = 32 is 3.25 rounds = 4 x
// 32; S = 36 is 2.9 rounds = 3 x 36: -16 %.
__host__ inline int ks_sliced_pick_sets(size_t count, int slices, int slots) {
  int best = kKsSlSets;
  size_t best_cost = ~(size_t)0;
  for (int sets : {24, 28, 32, 36, 40}) {
    const size_t grid = ((count + (size_t)ks_sliced_cts(sets) - 1) / (size_t)ks_sliced_cts(sets)) * (size_t)slices;
    const size_t cost = ((grid + (size_t)slots - 1) / (size_t)slots) * (size_t)sets;
    if (cost < best_cost || (cost == best_cost && sets == kKsSlSets)) {
      best = sets;
      best_cost = cost;
    }
  }
  return best;
}

template <int IC, int S = kKsSlSets>
__global__ __launch_bounds__(256) void k_key_switch_sliced(const uint32_t *__restrict__ lv1,  // [count][N+1]
                                                            const unsigned char *__restrict__ ksk,  // engine layout
                                                            int n, int basebit, int t,
                                                            uint32_t *__restrict__ out, size_t count) {
  // gridDim.z = K chunks: small batches cut the walk over the N coefficients into that many workgroups, which meet
  // in the (then zeroed) output through integer atomics -- u32 addition commutes: same bits in any arrival order
  constexpr int N = 1024, NS = kKsSlSlots, D = NS - 1, CTS = ks_sliced_cts(S);
  const int kchunks = (int)gridDim.z, i_begin = (int)blockIdx.z * (N / kchunks), i_end = i_begin + N / kchunks;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  extern __shared__ __attribute__((aligned(16))) unsigned char sl_smem[];
  const uint32_t base = 1u << basebit;
  const uint32_t row_bytes = (uint32_t)ksk_row_words(n) * 4u;
  const uint32_t slot_bytes = ks_sliced_slot_bytes((int)base);
  const uint32_t cw = slot_bytes >> 12;  // DMA instructions per wave per group (1 KiB = 4 row slices each)
  const uint32_t off_ab = NS * slot_bytes;
  uint32_t(*s_ab)[IC] = reinterpret_cast<uint32_t(*)[IC]>(sl_smem + off_ab);  // [ciphertext][coefficient]
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)sl_smem;

  const int tid = threadIdx.x;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane = (uint32_t)(tid & 63);
  const uint32_t sub = lane >> 4, c4 = lane & 15u;
  const size_t ct0 = (size_t)blockIdx.x * CTS;
  const uint32_t col0 = blockIdx.y * 64u;  // first column of this slice
  const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
  const uint32_t total = (uint32_t)N * (uint32_t)t;
  const uint32_t total_rows = total * base;

  // group q: rows q*base .. q*base+base-1; DMA instruction x of a group moves row slices 4x..4x+3
  // (lane quarter = row, 16 lanes x 16 B = the 256-byte slice).  Rows past the group (padding up to a
  // whole instruction) and past the key are clamped to a valid row: they land in LDS rows no digit selects.
  auto dma_group = [&](uint32_t q) {
    const uint32_t slot = q % NS;
    for (uint32_t c = 0; c < cw; ++c) {
      const uint32_t x = wave + c * kKsSlWaves;
      uint32_t row = (q < total ? q : total - 1u) * base + 4u * x + sub;
      row = row < total_rows ? row : total_rows - 1u;
      const size_t gofs = (size_t)row * row_bytes + col0 * 4u + c4 * 16u;
      glds16(ksk + gofs, lds_base + slot * slot_bytes + x * 1024u);
    }
  };

  u32x4 acc[S];
#pragma unroll
  for (int a = 0; a < S; ++a) acc[a] = u32x4{0u, 0u, 0u, 0u};

  uint32_t q = (uint32_t)i_begin * (uint32_t)t;
#pragma unroll 1
  for (uint32_t d = 0; d < (uint32_t)D; ++d) dma_group(q + d);

#pragma unroll 1
  for (int i0 = i_begin; i0 < i_end; i0 += IC) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int idx = tid; idx < CTS * IC; idx += 256) {
      const int c = idx / IC, ii = idx % IC;
      const size_t ct = ct0 + c;
      s_ab[c][ii] = ct < count ? lv1[ct * (N + 1) + i0 + ii] + prec_offset : 0u;
    }
    __syncthreads();
#pragma unroll 1
    for (int ii = 0; ii < IC; ++ii) {
      uint32_t ab[S];  // this lane quarter's ciphertexts: wave*4S + 4a + sub
#pragma unroll
      for (int a = 0; a < S; ++a) ab[a] = s_ab[wave * (4 * S) + 4 * a + sub][ii];
#pragma unroll 1
      for (int j = 0; j < t; ++j, ++q) {
        // every wave issues exactly cw DMAs per group: this wave's pieces of group q have landed when at
        // most the D-1 younger groups' are outstanding; the barrier makes all waves' pieces visible and
        // retires group q-1, whose slot the next DMA refills
        if (cw == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 1) : "memory");
        else if (cw == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 2) : "memory");
        else if (cw == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 4) : "memory");
        else if (cw == 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 8) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dma_group(q + (uint32_t)D);
        const uint32_t sh = 32u - (uint32_t)(j + 1) * (uint32_t)basebit;
        const uint32_t lane_base = (q % NS) * slot_bytes + c4 * 16u;
#ifndef TFHE_KS_SL_GB36
#define TFHE_KS_SL_GB36 6
#endif
        // LDS reads in flight per lane before their subtractions; divides S (24 .. 40 in steps of 4)
        constexpr int GB = S % 8 == 0 ? 8 : (S == 36 ? TFHE_KS_SL_GB36 : 4);
        static_assert(S % GB == 0, "the read group must divide S");
#pragma unroll
        for (int gb = 0; gb < S; gb += GB) {
          u32x4 v[GB];
#pragma unroll
          for (int g = 0; g < GB; ++g) {
            const uint32_t k = __builtin_amdgcn_ubfe(ab[gb + g], sh, (uint32_t)basebit);
            v[g] = *reinterpret_cast<const u32x4 *>(sl_smem + ((k << 8) + lane_base));
          }
#pragma unroll
          for (int g = 0; g < GB; ++g) acc[gb + g] -= v[g];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int a = 0; a < S; ++a) {
    const size_t ct = ct0 + wave * (4 * S) + 4 * a + sub;
    if (ct < count) {
      uint32_t *o = out + ct * (size_t)(n + 1);
      const uint32_t w[4] = {acc[a].x, acc[a].y, acc[a].z, acc[a].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int x = (int)(col0 + c4 * 4u) + c;
        if (kchunks == 1) {
          if (x < n) o[x] = w[c];
          if (x == n) o[x] = w[c] + lv1[ct * (N + 1) + N];  // res.b = src.b - sum (trgsw.rs:342)
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

}  // namespace tfhe
