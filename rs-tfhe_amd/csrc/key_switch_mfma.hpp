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
// Tiling: a workgroup of kKmWaves = 4 waves owns 32 * kKmFrags * 4 ciphertexts x one column block (the
// ceil((n+1)/32) output tiles are dealt to kKmColBlocks blocks as evenly as possible) x ONE byte plane; a wave owns
// 32 * kKmFrags rows (kKmFrags A fragments) of all NT tiles of the block: kKmFrags * NT accumulator tiles of 16
// registers.  Product defaults: one fragment, two column blocks (NT = 11 at n = 700: 176 accumulator registers,
// two workgroups per CU, two waves per SIMD); two fragments x four blocks halves the LDS reads and the L2 -> LDS
// traffic and measured 6 % slower (profiles/exp/logs/r3g_ab_ks_frags.log).  The four planes of an output word are
// merged with integer atomics.  One K-step is K = 32: 8 digit groups (i, j) x 4 candidate rows k.
//   * B (the key plane) is streamed global -> LDS by global_load_lds_dwordx4 into a 4-slot ring (one slot =
//     one K-step for the whole workgroup), three steps ahead, handed over by a counted s_waitcnt vmcnt + one
//     s_barrier per step; every wave reads all tiles of the slot (ds_read_b128, linear: conflict-free).  The
//     key is laid out at load time in exactly this fragment order (k_ksk_planes), so a tile is one contiguous KiB
//     and a workgroup's whole walk one contiguous stream.
//   * A (the one-hot) is never stored: a lane builds its 16 bytes of a fragment in registers from four
//     a_bar words, dword c = 1 << 8*digit.  The a_bar words of the wave's rows are staged 16 coefficients at a
//     time in a wave-private LDS buffer by dword DMAs issued one block ahead.
//   * The order of the K axis is free (it is a sum); it is chosen so that a lane's four dwords of a step
//     are the SAME digit position j of four consecutive coefficients: one ds_read_b128 and four
//     (add, shift, and, shift) per fragment per step.
// Bound: the matrix pipes and the clock they are allowed (operand-toggling limited: profiles/exp/ubench_mfma.hip).
// 65,536 x 704 x 36,864 x 4 planes x 2 = 1.36e13 int8 ops per launch at SECURITY_128_BIT (a quarter of them
// against the zero k = 0 rows); the key planes cross L2 -> LDS once per workgroup (52 GB per launch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "experiment.hpp"

namespace tfhe {

#ifndef TFHE_KM_FRAGS  // experiment knobs (profiles/exp/build_variants.sh); the defaults are the product
#define TFHE_KM_FRAGS 1
#endif
#ifndef TFHE_KM_COLBLOCKS
#define TFHE_KM_COLBLOCKS 2
#endif
constexpr int kKmWaves = 4;              // waves per workgroup (two workgroups per CU: 2 waves per SIMD)
constexpr int kKmFrags = TFHE_KM_FRAGS;  // 32-row A fragments per wave: a key tile read from LDS feeds this many matrix instructions
constexpr int kKmRows = 32 * kKmFrags * kKmWaves;  // ciphertexts per workgroup; level-1 buffers are padded to a multiple
#ifndef TFHE_KM_SLOTS
#define TFHE_KM_SLOTS 4
#endif
constexpr int kKmSlots = TFHE_KM_SLOTS;  // ring depth in K-steps
constexpr int kKmAhead = kKmSlots - 1;   // DMA lead in K-steps
constexpr int kKmAbBytes = 32 * kKmFrags * 16 * 4;  // one wave's a_bar stage: its rows x 16 coefficients
constexpr int kKmAbQ = kKmAbBytes / 256; // dword DMA instructions per stage (256 B each)
constexpr int kKmColBlocks = TFHE_KM_COLBLOCKS;
constexpr int kKmMaxTiles = 12 / kKmFrags;  // accumulator budget: kKmFrags x NT x 16 registers <= 192 (two waves per SIMD)

// key tiles a wave copies per K-step, tile positions per ring slot
__host__ __device__ constexpr int ks_mfma_tpw(int nt) { return (nt + kKmWaves - 1) / kKmWaves; }
__host__ __device__ constexpr int ks_mfma_slot_bytes(int nt) { return ks_mfma_tpw(nt) * kKmWaves * 1024; }
__host__ __device__ constexpr size_t ks_mfma_lds_bytes(int nt) {
  return (size_t)kKmSlots * ks_mfma_slot_bytes(nt) + (size_t)kKmWaves * 2 * kKmAbBytes + (size_t)kKmWaves * 256;
}
// The ceil((n+1)/32) column tiles are dealt to the column blocks as evenly as possible: the first `tiles % 4`
// blocks hold one more than the others.  ks_mfma_tiles = the larger count (the kernel's NT).
__host__ __device__ __forceinline__ int ks_mfma_total_tiles(int n) { return (n + 1 + 31) / 32; }
__host__ __device__ __forceinline__ int ks_mfma_tiles(int n) { return (ks_mfma_total_tiles(n) + kKmColBlocks - 1) / kKmColBlocks; }
__host__ __device__ __forceinline__ int ks_mfma_block_tiles(int n, int cb) {
  const int tiles = ks_mfma_total_tiles(n);
  return tiles / kKmColBlocks + (cb < tiles % kKmColBlocks ? 1 : 0);
}
__host__ __device__ __forceinline__ int ks_mfma_block_first(int n, int cb) {  // first tile of column block cb
  const int tiles = ks_mfma_total_tiles(n), base = tiles / kKmColBlocks, rem = tiles % kKmColBlocks;
  return cb * base + (cb < rem ? cb : rem);
}
__host__ __device__ __forceinline__ size_t ks_mfma_key_bytes(int n, int t) {
  return (size_t)4 * (64 * 2 * t) * ks_mfma_total_tiles(n) * 1024;
}
constexpr size_t kKmKeyTailPad = 64 * 1024;  // the DMA lead runs kKmAhead steps (<= 8 KiB each) past the last plane

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
//   [plane p][column block cb][K-step s][tile c < block tiles][lane][16 B],  s = blk*2t + 2j + hh  (blk: 16-coefficient
//   block, j: digit position, hh: which 8 coefficients), lane = (column in tile = lane & 31, kb = lane >> 5),
//   byte 4cc + k = plane byte of key row (i = 16 blk + 8 hh + 4 kb + cc, j, k) at that column.
__global__ void k_ksk_planes(const uint32_t *__restrict__ eng, unsigned char *__restrict__ out, int n, int t,
                             size_t chunks) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= chunks) return;
  const int lane = (int)(idx & 63);
  size_t r = idx >> 6;  // (p, cb, s, c) with a per-block tile count
  const int S = 64 * 2 * t, tiles = ks_mfma_total_tiles(n);
  const size_t per_plane = (size_t)S * tiles;
  const int p = (int)(r / per_plane);
  r -= (size_t)p * per_plane;
  int cb = 0;
  while (cb + 1 < kKmColBlocks && r >= (size_t)S * ks_mfma_block_first(n, cb + 1)) ++cb;
  r -= (size_t)S * ks_mfma_block_first(n, cb);
  const int nt = ks_mfma_block_tiles(n, cb);
  const int s = (int)(r / (size_t)nt), c = (int)(r % (size_t)nt);
  const int col = (ks_mfma_block_first(n, cb) + c) * 32 + (lane & 31), kb = lane >> 5;
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

// global -> LDS DMA: 64 lanes x 16 B (4 B) from sbase + voff to LDS address `lds` + lane x 16 (4); inline asm so that
// hipcc neither drains it with vmcnt(0) at the next LDS read nor loses track of M0 (saved and restored here)
__device__ __forceinline__ void km_dma16(uint32_t voff, const void *sbase, uint32_t lds) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds)
      : "memory");
}
__device__ __forceinline__ void km_dma4(uint32_t voff, const void *sbase, uint32_t lds) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds)
      : "memory");
}

// position in the K walk: 16-coefficient block, step within the block
struct KmPos {
  int blk, u;
};

// A workgroup = kKmWaves waves x kKmFrags x 32 rows, one column block, one byte plane, one chunk of the walk over K.
// lv1 must be readable for count rounded up to kKmRows rows (rows past count are computed and dropped).  out must be
// zero on entry: the four byte planes and the `ksplit` K chunks (both decoded from the linear workgroup index below)
// are merged with integer atomics (u32 addition commutes: same bits in any arrival order).
// NT = tiles of the widest column block; a block with NT-1 tiles skips the last tile (wave-uniform branch).
#ifndef TFHE_KM_XCD
#define TFHE_KM_XCD 1
#endif
template <int NT>
__global__ __launch_bounds__(64 * kKmWaves, 2) void k_key_switch_mfma(const uint32_t *__restrict__ lv1,        // [count][N+1]
                                                                       const unsigned char *__restrict__ ksk8,  // k_ksk_planes layout
                                                                       int n, int t, uint32_t *__restrict__ out,  // [count][n+1]
                                                                       size_t count,
                                                                       unsigned long long *clk,  // optional [2]: += shader cycles, += constant-rate ticks
                                                                       int ksplit) {  // 1, 2, 4, 8 or 16: the walk over the 64 coefficient blocks is cut into this many workgroups
  constexpr int N = 1024, D = kKmAhead, WAVES = kKmWaves, R = kKmFrags;
  constexpr int TPW = ks_mfma_tpw(NT), SLOT = ks_mfma_slot_bytes(NT), OPS = TPW + R;
  const unsigned long long clk0 = clk ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rtc0 = clk ? __builtin_amdgcn_s_memrealtime() : 0ull;
  extern __shared__ __attribute__((aligned(16))) unsigned char km_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)km_smem;
  const uint32_t off_ab = (uint32_t)(kKmSlots * SLOT) + (uint32_t)wave * 2u * kKmAbBytes;
  const uint32_t off_dump = (uint32_t)(kKmSlots * SLOT) + (uint32_t)(WAVES * 2 * kKmAbBytes) + (uint32_t)wave * 256u;
#if TFHE_KM_XCD
  // Workgroups are handed to the 8 XCDs round-robin in dispatch order (x fastest).  With 2 column blocks x 4 planes =
  // 8 (plane, column block) streams, giving workgroup `lin` stream lin % 8 puts each stream on ONE XCD: that XCD's L2
  // then holds one stream's window (all its row blocks walk K together) instead of a window of all eight.
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned combos = gridDim.y * gridDim.z, combo = lin % combos, rest = lin / combos;
  const int cb = (int)(combo % gridDim.y), plane = (int)(combo / gridDim.y);  // one byte plane per workgroup
#else
  const unsigned rest = blockIdx.x;
  const int cb = blockIdx.y, plane = blockIdx.z;  // one byte plane per workgroup: no epilogue inside the K loop
#endif
  // grid.x = row blocks x K chunks: small batches cut the walk over K so that the whole chip works on them (the
  // chunks, like the planes, meet in the output through integer atomics)
  const unsigned n_rb = gridDim.x / (unsigned)ksplit;
  const int kchunk = (int)(rest / n_rb);                 // this workgroup's K chunk
  const int blk0 = kchunk * (64 / ksplit);               // its first coefficient block (even: the a_bar double buffer starts at 0)
  const size_t row0 = (size_t)(rest % n_rb) * kKmRows + (size_t)wave * (32 * R);  // this wave's rows
  const int nt_blk = ks_mfma_block_tiles(n, cb), tile0 = ks_mfma_block_first(n, cb);
  const bool full = nt_blk == NT;  // wave-uniform
  const int spb = 2 * t, S = 64 * spb;  // steps per block, per plane
  const uint32_t prec = 1u << (31 - 2 * t);
  const uint32_t v16 = (uint32_t)lane * 16u;
  // a_bar DMA piece q of a block: rows 4q + (lane >> 4), coefficient 16 blk + (lane & 15)
  const uint32_t v4 = (uint32_t)(((lane >> 4) * (N + 1) + (lane & 15)) * 4);
  const uint32_t *ab_row0 = lv1 + row0 * (size_t)(N + 1);

  // The key tiles of this workgroup's (plane, column block) are one contiguous stream, nt_blk KiB per K-step: a DMA
  // source is the stream base (SGPRs) + a per-lane 32-bit offset that advances by nt_blk KiB per step.  The D steps
  // fetched past the end of the walk read the next block (or the allocation's tail pad) into slots nobody reads.
  const unsigned char *kplane = ksk8 + ((size_t)plane * ks_mfma_total_tiles(n) + (size_t)tile0) * S * 1024;
  uint32_t koff[TPW];  // this wave's tiles: wave, wave + WAVES, ... (clamped past nt_blk: lands in a position nobody reads)
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tile = wave + i * WAVES;
    koff[i] = (uint32_t)(tile < nt_blk ? tile : nt_blk - 1) * 1024u + v16;
  }
  const uint32_t kstride = (uint32_t)nt_blk * 1024u;
  const int steps = S / ksplit;  // K-steps of this workgroup: 64 / ksplit blocks
#pragma unroll
  for (int i = 0; i < TPW; ++i) koff[i] += (uint32_t)(blk0 * spb) * kstride;
  auto dma_key = [&](uint32_t slot) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      km_dma16(koff[i], kplane, slot + (uint32_t)(wave + i * WAVES) * 1024u);
      koff[i] += kstride;
    }
  };

  // ---- prologue: a_bar block 0, key steps 0 .. D-1 ---------------------------------------------
  for (int q = 0; q < kKmAbQ; ++q) km_dma4(v4, ab_row0 + (size_t)(4 * q) * (N + 1) + 16 * blk0, lds_base + off_ab + (uint32_t)q * 256u);
  for (int d = 0; d < D; ++d) dma_key(lds_base + (uint32_t)d * SLOT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  km_i32x16 acc[R][NT];
  const uint32_t a_lane = (uint32_t)((lane & 31) * 64 + (lane >> 5) * 16);  // this lane's 4 words in a stage row
  // A fragment f of step (blk, u): digit position j = u >> 1 of coefficients 16 blk + 8 (u & 1) + 4 kb + (0..3).
  // Reading the four a_bar words and building the one-hot bytes are separate so that the LDS latency of the read
  // can be covered by matrix instructions.
  auto read_a = [&](const KmPos &q, int f) -> km_u32x4 {
    const unsigned char *abuf = km_smem + off_ab + (uint32_t)(q.blk & 1) * kKmAbBytes + a_lane + (uint32_t)(q.u & 1) * 32u;
    return *reinterpret_cast<const km_u32x4 *>(abuf + f * (32 * 64));
  };
  auto build_a = [&](const KmPos &q, km_u32x4 w) -> km_i32x4 {
    const uint32_t sh = (uint32_t)(27 - 2 * (q.u >> 1));  // ((a_bar >> (30 - 2j)) & 3) * 8
    km_u32x4 a;
    a.x = 1u << (((w.x + prec) >> sh) & 0x18u);
    a.y = 1u << (((w.y + prec) >> sh) & 0x18u);
    a.z = 1u << (((w.z + prec) >> sh) & 0x18u);
    a.w = 1u << (((w.w + prec) >> sh) & 0x18u);
    return __builtin_bit_cast(km_i32x4, a);
  };
  auto make_a = [&](const KmPos &q, int f) -> km_i32x4 { return build_a(q, read_a(q, f)); };
  KmPos cur{blk0, 0};
  km_i32x4 A[R];
#pragma unroll
  for (int f = 0; f < R; ++f) A[f] = make_a(cur, f);
#ifndef TFHE_KM_PRE
#define TFHE_KM_PRE 3
#endif
  constexpr int PRE = NT < TFHE_KM_PRE ? NT : TFHE_KM_PRE;  // key tiles in flight ahead of the matrix instruction that consumes them
  constexpr int H = NT / 2;             // tiles consumed before the mid-step barrier
  km_i32x4 Bpre[PRE];
#pragma unroll
  for (int c = 0; c < PRE; ++c) Bpre[c] = *reinterpret_cast<const km_i32x4 *>(km_smem + v16 + c * 1024);
#pragma unroll
  for (int f = 0; f < R; ++f)
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[f][c] = km_i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // Step g consumes slot g % 4.  The hand-over of step g+1 sits in the MIDDLE of step g: by then my own pieces of
  // step g+1 have landed (one younger DMA group may be in flight), after the barrier everybody's have, and everybody
  // is done with step g-1, whose slot the next DMA group refills.  So the first key tiles of step g+1 and its A
  // fragments are fetched under the second half of step g's matrix instructions and no wave starts a step waiting.
  // The second half is interleaved BY HAND (sched_barrier): one matrix instruction, then one piece of the step's
  // other work (a DMA with its scalar address arithmetic, the next A fragment, ...), so that the ~60 non-matrix
  // instructions of a step issue inside the 32-cycle shadows of the matrix pipe instead of in one block between two
  // matrix instructions (what the scheduler does with the asm statements on its own).
  auto run = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;  // every tile of the block exists (else the last one is skipped)
#pragma unroll 1
    for (int g = 0; g < steps; ++g) {
      const unsigned char *slot = km_smem + (uint32_t)(g % kKmSlots) * SLOT + v16;
      const unsigned char *slot_next = km_smem + (uint32_t)((g + 1) % kKmSlots) * SLOT + v16;
      km_i32x4 B[NT];
#pragma unroll
      for (int c = 0; c < PRE; ++c) B[c] = Bpre[c];
#pragma unroll
      for (int c = 0; c < H; ++c) {
        if (c + PRE < NT) B[c + PRE] = *reinterpret_cast<const km_i32x4 *>(slot + (c + PRE) * 1024);
#pragma unroll
        for (int f = 0; f < R; ++f) acc[f][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[f], B[c], acc[f][c], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // the first half's matrix instructions stay ahead of the barrier
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * OPS) : "memory");
      if (!TFHE_ABL_KM_NOBARRIER) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const uint32_t dslot = lds_base + (uint32_t)((g + D) % kKmSlots) * SLOT;
      const int nblk = (cur.blk + 1) & 63;
      const uint32_t nbuf = lds_base + off_ab + (uint32_t)((cur.blk + 1) & 1) * kKmAbBytes;
      const int u_now = cur.u;
      constexpr int PIECES = TPW + R + 2;
      km_i32x4 A_next[R];
      km_u32x4 Aw[R];
      // piece `it` of the step's other work, one per matrix instruction of the second half.  First the LDS reads the
      // NEXT step starts from (its a_bar words and first key tiles: valid after the barrier above) so that their
      // latency lies under this step's remaining matrix instructions; then the TPW key DMAs and the R a_bar DMAs
      // (pieces R*u .. R*u + R-1 of the next block, real while < kKmAbQ; the last real one is issued at step 7 of a
      // block, >= D steps before that block's first fragment is built), with the one-hot bytes of the next A
      // fragment as the second-to-last piece (so that only a short DMA trails the step's last matrix instruction).
      auto piece = [&](int it) {
        if (it == 0) {
          if (++cur.u == spb) {
            cur.u = 0;
            cur.blk = (cur.blk + 1) & 63;
          }
#pragma unroll
          for (int f = 0; f < R; ++f) Aw[f] = read_a(cur, f);
#pragma unroll
          for (int c = 0; c < PRE; ++c) Bpre[c] = *reinterpret_cast<const km_i32x4 *>(slot_next + c * 1024);
        } else if (it == PIECES - 2) {  // second to last: the one-hot bytes, still inside a matrix shadow
#pragma unroll
          for (int f = 0; f < R; ++f) A_next[f] = build_a(cur, Aw[f]);
        } else {  // the DMAs, in order: TPW key tiles, R a_bar pieces
          const int d = it - 1 - (it > PIECES - 2 ? 1 : 0);
          if (d < TPW) {
            km_dma16(koff[d], kplane, dslot + (uint32_t)(wave + d * WAVES) * 1024u);
            koff[d] += kstride;
          } else {
            const int q0 = R * u_now + (d - TPW);
            const uint32_t *s0 = ab_row0 + (size_t)(4 * (q0 & (kKmAbQ - 1))) * (N + 1) + 16 * nblk;
            km_dma4(v4, s0, q0 < kKmAbQ ? nbuf + (uint32_t)q0 * 256u : lds_base + off_dump);
          }
        }
      };
#pragma unroll
      for (int c = H; c < NT; ++c) {
        if (c + PRE < NT) B[c + PRE] = *reinterpret_cast<const km_i32x4 *>(slot + (c + PRE) * 1024);
        if (c < NT - 1 || FULL) {
#pragma unroll
          for (int f = 0; f < R; ++f) acc[f][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[f], B[c], acc[f][c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c - H < PIECES) piece(c - H);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int it = NT - H; it < PIECES; ++it) piece(it);  // blocks with fewer tiles than pieces: the rest in a row
#pragma unroll
      for (int f = 0; f < R; ++f) A[f] = A_next[f];
    }
  };
  if (full)
    run(std::true_type{});
  else
    run(std::false_type{});
  // ---- merge this plane into the output ---------------------------------------------------------------
  // C tile element e of lane: row 32f + (e&3) + 8(e>>2) + 4(lane>>5), column 32c + (lane&31).  An address is one
  // per-lane offset + a wave-uniform one.
  {
    const int sh8 = 8 * plane;
    const int row_lim = (int)(count > row0 ? (count - row0 < 32 * R ? count - row0 : 32 * R) : 0) - 4 * (lane >> 5);  // valid e-rows: < row_lim
    const int col_lane = tile0 * 32 + (lane & 31);
    const uint32_t lane_off = (uint32_t)((4 * (lane >> 5)) * (n + 1) + col_lane) * 4u;
    unsigned char *obase = reinterpret_cast<unsigned char *>(out + row0 * (size_t)(n + 1));
    const uint32_t row_bytes = (uint32_t)(n + 1) * 4u;
#pragma unroll
    for (int f = 0; f < R; ++f)
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const bool col_ok = col_lane + 32 * c <= n && c < nt_blk;
        const bool is_body = col_lane + 32 * c == n && plane == 0 && kchunk == 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int er = 32 * f + (e & 3) + 8 * (e >> 2);
          if (col_ok && er < row_lim) {
            uint32_t v = 0u - ((uint32_t)acc[f][c][e] << sh8);
            if (is_body) v += lv1[(row0 + (size_t)(er + 4 * (lane >> 5))) * (size_t)(N + 1) + N];  // res.b = src.b - sum (trgsw.rs:342)
            unsigned char *uni = obase + (size_t)((uint32_t)er * row_bytes + (uint32_t)(c * 128));
            if (v) atomicAdd(reinterpret_cast<uint32_t *>(uni + lane_off), v);
          }
        }
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing DMAs before the LDS goes away
  if (clk && tid == 0) {
    atomicAdd(&clk[0], __builtin_amdgcn_s_memtime() - clk0);
    atomicAdd(&clk[1], __builtin_amdgcn_s_memrealtime() - rtc0);
  }
}

}  // namespace tfhe
