// key_switch_mfma.hpp -- identity key switching at base 4 as an exact int8 matrix product.
//
// identity_key_switching (src/trgsw.rs:332-360) over a batch is a contraction:
//
//   out[b][x] = src.b[b]*(x == n) - sum_{i<N, j<t} KSK[i][j][digit(b, i, j)][x]      (mod 2^32)
//             = src.b[b]*(x == n) - ( OneHot[b][(i,j,k)] . KSK[(i,j,k)][x] )
//
// with OneHot[b][(i,j,k)] = 1 iff k == digit(b, i, j) (k = 0 rows of the key are zero, key.rs:107-118).
// The u32 key words are split ONCE into four balanced signed byte planes, w = sum_p s_p * 256^p (mod 2^32),
// s_p in [-128, 127], so each plane is an i8 x i8 -> i32 product on the matrix cores
// (v_mfma_i32_32x32x32_i8): |acc_p| <= N*t*128 < 2^21, no overflow, and the planes recombine as
// sum_p acc_p << 8p with wrapping u32 arithmetic -- bit-exact, not approximately equal.
//
// Tiling: a workgroup of 4 waves owns 256 ciphertexts x one column block (half of the n+1 output words,
// NT tiles of 32 columns); wave w owns rows 64w .. 64w+63 (two 32-row A fragments) of all NT tiles:
// 2*NT accumulator tiles of 16 registers (352 at n = 700), one wave per SIMD.  One K-step is K = 32:
// 8 digit groups (i, j) x 4 candidate rows k.
//   * B (the key plane) is streamed global -> LDS by global_load_lds_dwordx4 into a 4-slot ring (one slot =
//     NT KiB = one K-step for the whole workgroup), three steps ahead, handed over by a counted
//     s_waitcnt vmcnt + one s_barrier per step; every wave reads all NT tiles of the slot (ds_read_b128,
//     linear: conflict-free).  The key is laid out at load time in exactly this fragment order
//     (k_ksk_planes), so a tile is one contiguous KiB.
//   * A (the one-hot) is never stored: a lane builds its 16 bytes of a fragment in registers from four
//     a_bar words, dword c = 1 << 8*digit.  The a_bar words of the wave's 64 rows are staged 16
//     coefficients at a time in a wave-private LDS buffer by dword DMAs issued one block ahead.
//   * The order of the K axis is free (it is a sum); it is chosen so that a lane's four dwords of a step
//     are the SAME digit position j of four consecutive coefficients: one ds_read_b128 and four
//     (add, shift, and, shift) per fragment per step.
// Bound: the matrix pipe.  65,536 x 704 x 32,768 x 4 planes x 2 = 1.21e16 int8 ops per launch at
// SECURITY_128_BIT; the key planes cross L2 -> LDS once per workgroup (23.6 GB per launch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfhe {

constexpr int kKmRows = 256;             // ciphertexts per workgroup
constexpr int kKmWaves = 8;              // 32 rows each: 2 waves per SIMD, 16*NT accumulator registers per wave
constexpr int kKmSlots = 4;              // ring depth in K-steps
constexpr int kKmAhead = kKmSlots - 1;   // DMA lead in K-steps
constexpr int kKmSlotTiles = 16;         // tile positions per slot (2 per wave)
constexpr int kKmSlotBytes = kKmSlotTiles * 1024;
constexpr int kKmAbBytes = 32 * 16 * 4;  // one wave's a_bar stage: 32 rows x 16 coefficients
constexpr int kKmAbQ = kKmAbBytes / 256; // dword DMA instructions per stage (256 B each)
constexpr int kKmOpsPerStep = 3;         // per wave per K-step: 2 key tiles + 1 a_bar piece
constexpr int kKmColBlocks = 2;

__host__ __device__ __forceinline__ size_t ks_mfma_lds_bytes() {
  return (size_t)kKmSlots * kKmSlotBytes + kKmWaves * 2 * kKmAbBytes + kKmWaves * 256;
}
// 32-column tiles per column block for n+1 output words
__host__ __device__ __forceinline__ int ks_mfma_tiles(int n) {
  const int tiles = (n + 1 + 31) / 32;
  return (tiles + kKmColBlocks - 1) / kKmColBlocks;
}
__host__ __device__ __forceinline__ size_t ks_mfma_key_bytes(int n, int t, int nt) {
  return (size_t)4 * kKmColBlocks * (64 * 2 * t) * nt * 1024;
}

// Balanced signed byte p of w: w = sum_p s_p 256^p (mod 2^32), s_p in [-128, 127].
__host__ __device__ __forceinline__ uint32_t ks_plane_byte(uint32_t w, int p) {
  uint32_t s = 0;
  for (int q = 0; q <= p; ++q) {
    s = w & 0xFFu;
    const uint32_t sext = (s & 0x80u) ? (s | 0xFFFFFF00u) : s;
    w = (w - sext) >> 8;
  }
  return s;
}

// u32 engine layout [N*t*4][RW] -> byte planes in MFMA fragment order:
//   [plane p][column block cb][K-step s][tile c][lane][16 B],  s = blk*2t + 2j + hh  (blk: 16-coefficient block,
//   j: digit position, hh: which 8 coefficients), lane = (column in tile = lane & 31, kb = lane >> 5),
//   byte 4cc + k = plane byte of key row (i = 16 blk + 8 hh + 4 kb + cc, j, k) at that column.
__global__ void k_ksk_planes(const uint32_t *__restrict__ eng, unsigned char *__restrict__ out, int n, int t, int nt,
                             size_t chunks) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= chunks) return;
  const int lane = (int)(idx & 63);
  size_t r = idx >> 6;
  const int c = (int)(r % (size_t)nt);
  r /= (size_t)nt;
  const int S = 64 * 2 * t;
  const int s = (int)(r % (size_t)S);
  r /= (size_t)S;
  const int cb = (int)(r % kKmColBlocks), p = (int)(r / kKmColBlocks);
  const int col = (cb * nt + c) * 32 + (lane & 31), kb = lane >> 5;
  const int blk = s / (2 * t), u = s % (2 * t), j = u >> 1, hh = u & 1;
  const int rw = (n + 1 + 3) & ~3;
  uint32_t o[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const int i = 16 * blk + 8 * hh + 4 * kb + cc;
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t row = ((size_t)i * t + j) * 4 + k;
      const uint32_t w = (col <= n && k != 0) ? eng[row * (size_t)rw + col] : 0u;
      d |= ks_plane_byte(w, p) << (8 * k);
    }
    o[cc] = d;
  }
  reinterpret_cast<uint4 *>(out)[idx] = make_uint4(o[0], o[1], o[2], o[3]);
}

using km_i32x4 = __attribute__((ext_vector_type(4))) int;
using km_i32x16 = __attribute__((ext_vector_type(16))) int;
using km_u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// One K-step's DMAs of one wave, as ONE asm statement (M0 is saved and restored around it; hipcc does not
// model M0 as clobberable).  Two 1-KiB pieces of the key tile row + one 256-byte piece of a_bar words.
__device__ __forceinline__ void km_dma_step(uint32_t v16, const void *b0, const void *b1, uint32_t l0, uint32_t l1,
                                            uint32_t v4, const void *a0, uint32_t la0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "s_mov_b32 m0, %8\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %6, %7\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v16), "s"(b0), "s"(b1), "s"(l0), "s"(l1), "v"(v4), "s"(a0), "s"(la0)
      : "memory");
}
__device__ __forceinline__ void km_dma_b2(uint32_t v16, const void *b0, const void *b1, uint32_t l0, uint32_t l1) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v16), "s"(b0), "s"(b1), "s"(l0), "s"(l1)
      : "memory");
}
__device__ __forceinline__ void km_dma_a1(uint32_t v4, const void *a0, uint32_t la0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v4), "s"(a0), "s"(la0)
      : "memory");
}

// position in the K walk: 16-coefficient block, step within the block
struct KmPos {
  int blk, u;
};

// lv1 must be readable for count rounded up to kKmRows rows (rows past count are computed and dropped).
// out must be zero on entry: the four byte planes (blockIdx.z) are merged with integer atomics (u32 addition
// commutes: same bits in any arrival order).
template <int NT>
__global__ __launch_bounds__(64 * kKmWaves, 2) void k_key_switch_mfma(const uint32_t *__restrict__ lv1,        // [count][N+1]
                                                                       const unsigned char *__restrict__ ksk8,  // k_ksk_planes layout
                                                                       int n, int t, uint32_t *__restrict__ out,  // [count][n+1]
                                                                       size_t count) {
  constexpr int N = 1024, D = kKmAhead;
  static_assert(NT <= kKmSlotTiles, "slot too small");
  extern __shared__ __attribute__((aligned(16))) unsigned char km_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)km_smem;
  const uint32_t off_ab = (uint32_t)(kKmSlots * kKmSlotBytes) + (uint32_t)wave * 2u * kKmAbBytes;
  const uint32_t off_dump = (uint32_t)(kKmSlots * kKmSlotBytes) + (uint32_t)(kKmWaves * 2 * kKmAbBytes) + (uint32_t)wave * 256u;
  const size_t row0 = (size_t)blockIdx.x * kKmRows + (size_t)wave * 32;  // this wave's 32 rows
  const int cb = blockIdx.y, plane = blockIdx.z;  // one byte plane per workgroup: no epilogue inside the K loop
  const int spb = 2 * t, S = 64 * spb;  // steps per block, per plane
  const uint32_t prec = 1u << (31 - 2 * t);
  const uint32_t v16 = (uint32_t)lane * 16u;
  // a_bar DMA op q of a block: rows 4q + (lane >> 4), coefficient 16 blk + (lane & 15)
  const uint32_t v4 = (uint32_t)(((lane >> 4) * (N + 1) + (lane & 15)) * 4);
  const uint32_t *ab_row0 = lv1 + row0 * (size_t)(N + 1);
  // key tiles this wave copies per step: wave, wave + 8 (clamped past NT: lands in a position nobody reads)
  const int tc0 = wave < NT ? wave : NT - 1, tc1 = wave + 8 < NT ? wave + 8 : NT - 1;

  // (past the last step the walk wraps to step 0: a harmless re-read into a slot nobody reads any more)
  const unsigned char *kplane = ksk8 + (size_t)(plane * kKmColBlocks + cb) * S * ((size_t)NT * 1024);
  auto key_step = [&](const KmPos &q) -> const unsigned char * {
    return kplane + (size_t)(q.blk * spb + q.u) * ((size_t)NT * 1024);
  };
  auto advance = [&](KmPos &q) {
    if (++q.u == spb) {
      q.u = 0;
      if (++q.blk == 64) q.blk = 0;
    }
  };

  // ---- prologue: a_bar block 0, key steps 0 .. D-1 ---------------------------------------------
  for (int q = 0; q < kKmAbQ; ++q) km_dma_a1(v4, ab_row0 + (size_t)(4 * q) * (N + 1), lds_base + off_ab + (uint32_t)q * 256u);
  KmPos pf{0, 0};  // next step to prefetch
  for (int d = 0; d < D; ++d) {
    const unsigned char *kb = key_step(pf);
    const uint32_t slot = lds_base + (uint32_t)d * kKmSlotBytes;
    km_dma_b2(v16, kb + tc0 * 1024, kb + tc1 * 1024, slot + (uint32_t)wave * 1024u, slot + (uint32_t)(wave + 8) * 1024u);
    advance(pf);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  km_i32x16 acc[NT];
  KmPos cur{0, 0};
  const uint32_t a_lane = (uint32_t)((lane & 31) * 64 + (lane >> 5) * 16);  // this lane's 4 words in a stage row
#pragma unroll
  for (int c = 0; c < NT; ++c) acc[c] = km_i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
  for (int g = 0; g < S; ++g) {
    // my pieces of step g have landed (the D-1 younger groups may be in flight); after the barrier everybody's
    // have, and everybody is done reading the slot of step g-1, which the next DMA group refills
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * kKmOpsPerStep) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      const unsigned char *kb = key_step(pf);
      const uint32_t slot = lds_base + (uint32_t)((g + D) % kKmSlots) * kKmSlotBytes;
      // a_bar words of the NEXT block into the other stage buffer: op u of the block (real while < kKmAbQ)
      const int nblk = (cur.blk + 1) & 63;
      const uint32_t nbuf = lds_base + off_ab + (uint32_t)((cur.blk + 1) & 1) * kKmAbBytes;
      const int q0 = cur.u;
      const uint32_t *s0 = ab_row0 + (size_t)(4 * (q0 & (kKmAbQ - 1))) * (N + 1) + 16 * nblk;
      const uint32_t d0 = q0 < kKmAbQ ? nbuf + (uint32_t)q0 * 256u : lds_base + off_dump;
      km_dma_step(v16, kb + tc0 * 1024, kb + tc1 * 1024, slot + (uint32_t)wave * 1024u, slot + (uint32_t)(wave + 8) * 1024u,
                  v4, s0, d0);
      advance(pf);
    }
    // ---- A fragment of this step: digit position j of coefficients 16 blk + 8 hh + 4 kb + (0..3) ----
    const int j = cur.u >> 1, hh = cur.u & 1;
    const uint32_t sh = (uint32_t)(27 - 2 * j);  // ((a_bar >> (30 - 2j)) & 3) * 8
    const unsigned char *abuf = km_smem + off_ab + (uint32_t)(cur.blk & 1) * kKmAbBytes + a_lane + (uint32_t)hh * 32u;
    const km_u32x4 w = *reinterpret_cast<const km_u32x4 *>(abuf);
    km_u32x4 a;
    a.x = 1u << (((w.x + prec) >> sh) & 0x18u);
    a.y = 1u << (((w.y + prec) >> sh) & 0x18u);
    a.z = 1u << (((w.z + prec) >> sh) & 0x18u);
    a.w = 1u << (((w.w + prec) >> sh) & 0x18u);
    const km_i32x4 A = __builtin_bit_cast(km_i32x4, a);
    const unsigned char *slot = km_smem + (uint32_t)(g % kKmSlots) * kKmSlotBytes + v16;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const km_i32x4 B = *reinterpret_cast<const km_i32x4 *>(slot + c * 1024);
      acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[c], 0, 0, 0);
    }
    if (++cur.u == spb) {
      cur.u = 0;
      cur.blk = (cur.blk + 1) & 63;
    }
  }
  // ---- merge this plane into the output ---------------------------------------------------------------
  // C tile element e of lane: row (e&3) + 8(e>>2) + 4(lane>>5), column 32c + (lane&31).  An address is one
  // per-lane offset + a wave-uniform one.
  {
    const int sh8 = 8 * plane;
    const int row_lim = (int)(count > row0 ? (count - row0 < 32 ? count - row0 : 32) : 0) - 4 * (lane >> 5);  // valid e-rows: < row_lim
    const int col_lane = cb * NT * 32 + (lane & 31);
    const uint32_t lane_off = (uint32_t)((4 * (lane >> 5)) * (n + 1) + col_lane) * 4u;
    unsigned char *obase = reinterpret_cast<unsigned char *>(out + row0 * (size_t)(n + 1));
    const uint32_t row_bytes = (uint32_t)(n + 1) * 4u;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const bool col_ok = col_lane + 32 * c <= n;
      const bool is_body = col_lane + 32 * c == n && plane == 0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int er = (e & 3) + 8 * (e >> 2);
        if (col_ok && er < row_lim) {
          uint32_t v = 0u - ((uint32_t)acc[c][e] << sh8);
          if (is_body) v += lv1[(row0 + (size_t)(er + 4 * (lane >> 5))) * (size_t)(N + 1) + N];  // res.b = src.b - sum (trgsw.rs:342)
          unsigned char *uni = obase + (size_t)((uint32_t)er * row_bytes + (uint32_t)(c * 128));
          if (v) atomicAdd(reinterpret_cast<uint32_t *>(uni + lane_off), v);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (wrapped) DMAs before the LDS goes away
}

}  // namespace tfhe
