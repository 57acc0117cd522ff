// ubench_tp.hip -- transpose A of the wave-resident FFT (fft512.hpp) through the LDS in two encodings (not product code).
//   hipcc --offload-arch=gfx950 -O2 -o profiles/exp/ubench_tp profiles/exp/ubench_tp.hip && ./profiles/exp/ubench_tp
// VERDICT r3 task 5(a): would `ds_write_addtid_b32` planes (4 B per lane, 2 cycles, no address VGPR) +
// `ds_read2st64_b32` gathers beat the shipped ds_write_b128 / ds_read_b128 transposes?  W waves of one workgroup
// (8 = two per SIMD, as the batch blind rotation runs) each repeat { F v_fma_f64 on the data, one transpose A (64 lanes x
// 8 complex) }; cycles per repetition from s_memtime.  Both forms are checked to realise the same permutation.
//   mode 0: the shipped form: 8 x ds_write_b128 (k*72 + lane), 8 x ds_read_b128 (hi*72 + lo + 8s)           16 LDS instructions
//   mode 1: planar: 32 x ds_write_addtid_b32 into four dword planes, 16 x ds_read2st64_b32 (lo / hi words)   48 LDS instructions
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <utility>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x2 = __attribute__((ext_vector_type(2))) int;

template <int OFF>
__device__ __forceinline__ void addtid(int v) {  // LDS[m0 + OFF + 4 * lane] = v
  asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(v), "n"(OFF) : "memory");
}
template <int K>
__device__ __forceinline__ void write_planes(const double (&re)[8], const double (&im)[8]) {
  addtid<0 * 2304 + K * 288>(__double2loint(re[K]));
  addtid<1 * 2304 + K * 288>(__double2hiint(re[K]));
  addtid<2 * 2304 + K * 288>(__double2loint(im[K]));
  addtid<3 * 2304 + K * 288>(__double2hiint(im[K]));
  if constexpr (K < 7) write_planes<K + 1>(re, im);
}

template <int MODE, int F>
__global__ __launch_bounds__(512) void k(unsigned long long *out, double *sink, int iters) {
  __shared__ __attribute__((aligned(256))) char lds[8 * 9216];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 3, lo = lane & 7;
  const unsigned tile = (unsigned)(size_t)lds + wave * 9216;
  double re[8], im[8], f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    re[i] = 1.0 + lane + 64 * i;  // element (slot i, lane)
    im[i] = -(1.0 + lane + 64 * i);
    f[i] = F ? 1.0 + 1e-9 * (lane + i) : 1.0;
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < F / 16; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        re[i] = __builtin_fma(re[i], f[i], 0.0);
        im[i] = __builtin_fma(im[i], f[(i + 1) & 7], 0.0);
      }
    if (MODE == 0) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        i32x4 v = {__double2loint(re[kk]), __double2hiint(re[kk]), __double2loint(im[kk]), __double2hiint(im[kk])};
        asm volatile("ds_write_b128 %0, %1" ::"v"(tile + (kk * 72 + lane) * 16), "v"(v) : "memory");
      }
      i32x4 v[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) asm volatile("ds_read_b128 %0, %1" : "=v"(v[s]) : "v"(tile + (hi * 72 + lo + 8 * s) * 16) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        re[s] = __hiloint2double(v[s].y, v[s].x);
        im[s] = __hiloint2double(v[s].w, v[s].z);
      }
    } else {
      // planes re.lo | re.hi | im.lo | im.hi: 8 rows of 72 dwords each = 2304 B = 9 x 256 B apart
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0" : "=&s"(keep) : "s"(__builtin_amdgcn_readfirstlane(tile)) : "memory");
      write_planes<0>(re, im);
      asm volatile("s_mov_b32 m0, %0" ::"s"(keep) : "memory");
      i32x2 a[8], b[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const unsigned addr = tile + (hi * 72 + lo + 8 * s) * 4;
        asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:9" : "=v"(a[s]) : "v"(addr) : "memory");
        asm volatile("ds_read2st64_b32 %0, %1 offset0:18 offset1:27" : "=v"(b[s]) : "v"(addr) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        re[s] = __hiloint2double(a[s].y, a[s].x);
        im[s] = __hiloint2double(b[s].y, b[s].x);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += re[i] * (i + 1) - im[i];
  sink[blockIdx.x * 512 + threadIdx.x] = acc;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE, int F>
double run(const char *name, unsigned long long *d_out, double *d_sink, int waves, double *checksum) {
  const int iters = 1000;
  hipLaunchKernelGGL((k<MODE, F>), dim3(1), dim3(64 * waves), 0, 0, d_out, d_sink, 1);  // one transpose: the permutation check
  CK(hipDeviceSynchronize());
  double h[512];
  CK(hipMemcpy(h, d_sink, sizeof(h), hipMemcpyDeviceToHost));
  *checksum = 0;
  for (int i = 0; i < 64; ++i) *checksum += h[i] * (i + 1);
  hipLaunchKernelGGL((k<MODE, F>), dim3(1), dim3(64 * waves), 0, 0, d_out, d_sink, iters);
  hipLaunchKernelGGL((k<MODE, F>), dim3(1), dim3(64 * waves), 0, 0, d_out, d_sink, iters);
  CK(hipDeviceSynchronize());
  unsigned long long c;
  CK(hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost));
  printf("  %-58s F=%3d %d waves: %8.1f cycles per {F fma + transpose} per wave\n", name, F, waves, (double)c / iters);
  return (double)c / iters;
}

int main() {
  unsigned long long *d;
  double *s;
  CK(hipMalloc(&d, 64));
  CK(hipMalloc(&s, 512 * 8));
  double c0, c1;
  for (int waves : {1, 4, 8}) {
    run<0, 0>("b128 write / b128 read (shipped)", d, s, waves, &c0);
    run<1, 0>("addtid_b32 planes / read2st64_b32", d, s, waves, &c1);
    printf("    same permutation: %s\n", c0 == c1 ? "yes" : "NO");
    run<0, 144>("b128 write / b128 read (shipped)", d, s, waves, &c0);
    run<1, 144>("addtid_b32 planes / read2st64_b32", d, s, waves, &c1);
    run<0, 288>("b128 write / b128 read (shipped)", d, s, waves, &c0);
    run<1, 288>("addtid_b32 planes / read2st64_b32", d, s, waves, &c1);
  }
  return 0;
}
