// ubench_xcu.hip -- what ONE cross-CU exchange of a spectrum costs (not product code).
//   hipcc --offload-arch=gfx950 -O2 -o profiles/exp/ubench_xcu profiles/exp/ubench_xcu.hip && ./profiles/exp/ubench_xcu
//
// The round-4 verdict's bounded latency experiment: split ONE ciphertext's CMUX step over two CUs of an XCD (the a-half
// rows on one, the b-half rows on the other), which needs one 8-KiB partial-spectrum exchange in each direction per
// step.  The phase ablations of k_blind_rotate_wide2 say what the split can save per step -- half the
// forward phase (2,500 -> ~1,250 cycles), half the digit preparation and multiply (~450) -- so the exchange has to cost
// less than ~1,700 - 0.15 x 7,290 = ~600 cycles (0.25 us) for the split to be 15 % faster.  This program measures the
// exchange itself, in the kernel's shape: two 512-thread workgroups on the SAME XCD (blocks b and b + 8: block b runs
// on XCD b % 8; checked with HW_REG_XCC_ID), each publishing 8 KiB (one 16-byte store per thread) and consuming the
// partner's 8 KiB, `iters` times back to back, every word checked.
//   mode 0  sc1 write-through stores -> s_waitcnt vmcnt(0) -> sc1 flag store; consumer: sc1 poll, sc1 loads
//           (the guide's R1 form: no fence, the L2 the two CUs share is the meeting point)
//   mode 1  plain stores -> __syncthreads -> lane-0 agent release fence -> flag; consumer: poll, agent acquire fence, plain loads
//   mode 2  flag only (no payload): the floor of any exchange
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void store_sc1(u32x4 *p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u32x4 load_sc1(const u32x4 *p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// buf: [pairs][2 sides][512] u32x4 payload; flag: [pairs][2] u32 (the iteration each side has published)
template <int MODE>
__global__ __launch_bounds__(512) void k(u32x4 *buf, unsigned *flag, unsigned long long *out, int iters) {
  const int pair = blockIdx.x & 7, side = blockIdx.x >> 3;  // blocks b and b + 8 land on XCD b % 8
  u32x4 *mine = buf + ((size_t)pair * 2 + side) * 512, *theirs = buf + ((size_t)pair * 2 + (side ^ 1)) * 512;
  unsigned *myflag = flag + pair * 2 + side, *theirflag = flag + pair * 2 + (side ^ 1);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int tid = threadIdx.x;
  unsigned long long errors = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 1; it <= iters; ++it) {
    if (MODE != 2) {
      const u32x4 v = {(unsigned)it, (unsigned)tid, (unsigned)side, (unsigned)(it * 2654435761u + tid)};
      if (MODE == 0) store_sc1(mine + tid, v);
      else mine[tid] = v;
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every thread's payload has left
    if (tid == 0) {
      if (MODE == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_store(myflag, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (long spin = 0; __hip_atomic_load(theirflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it; ++spin) {
        __builtin_amdgcn_s_sleep(1);
        if (spin > 20000000L) {  // (a partner that never came: give up instead of hanging the box)
          atomicAdd(&out[3], 1ull);
          break;
        }
      }
      if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();  // the partner's payload of this iteration is published
    if (MODE != 2) {
      const u32x4 w = MODE == 0 ? load_sc1(theirs + tid) : theirs[tid];
      if (w.x != (unsigned)it || w.y != (unsigned)tid || w.z != (unsigned)(side ^ 1) || w.w != (unsigned)(it * 2654435761u + tid)) ++errors;
    }
    // (the next iteration overwrites `mine`, which the partner may still be reading: it has to have consumed iteration it
    // before it can publish it + 1, and this side waits for THAT flag before it reads -- but not before it writes.  Two
    // buffers would be needed in a real kernel; here a second flag round closes the window.)
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(myflag + 16, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (long spin = 0; __hip_atomic_load(theirflag + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it; ++spin) {
        __builtin_amdgcn_s_sleep(1);
        if (spin > 20000000L) {
          atomicAdd(&out[3], 1ull);
          break;
        }
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (errors) atomicAdd(&out[2], errors);
  if (tid == 0) {
    atomicMax(&out[0], t1 - t0);
    atomicMax(&out[1], r1 - r0);
    out[8 + blockIdx.x] = xcc;
  }
}

template <int MODE>
void run(const char *name, u32x4 *buf, unsigned *flag, unsigned long long *out, int pairs) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(flag, 0, 64 * 4));
    CK(hipMemset(out, 0, 64 * 8));
    hipLaunchKernelGGL(k<MODE>, dim3(pairs == 8 ? 16 : 9), dim3(512), 0, 0, buf, flag, out, iters);
    CK(hipDeviceSynchronize());
  }
  unsigned long long h[64];
  CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  int same = 0;
  for (int p = 0; p < 8; ++p) same += h[8 + p] == h[8 + p + 8];
  // each iteration = one payload exchange + one flag-only round (see the kernel): report both, and their difference
  printf("%-44s %2d pairs  %8.0f shader cycles, %7.3f us per iteration (100 MHz counter)  errors %llu  timeouts %llu  pairs on one XCD: %d/8\n", name, pairs,
         (double)h[0] / iters, (double)h[1] / iters / 100.0, h[2], h[3], same);
}

int main() {
  u32x4 *buf;
  unsigned *flag;
  unsigned long long *out;
  CK(hipMalloc(&buf, 8 * 2 * 512 * 16));
  CK(hipMalloc(&flag, 64 * 4));
  CK(hipMalloc(&out, 64 * 8));
  CK(hipMemset(buf, 0, 8 * 2 * 512 * 16));
  for (int pairs : {8}) {
    run<2>("flag round only (x2 per iteration)", buf, flag, out, pairs);
    run<0>("sc1 stores + drained sc1 flag, sc1 loads", buf, flag, out, pairs);
    run<1>("plain stores + agent release / acquire", buf, flag, out, pairs);
  }
  return 0;
}
