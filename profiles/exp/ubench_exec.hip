// ubench_exec.hip -- does a v_fma_f64 with a partly empty EXEC mask issue faster?  (not product code)
//   hipcc --offload-arch=gfx950 -O2 -o profiles/exp/ubench_exec profiles/exp/ubench_exec.hip && ./profiles/exp/ubench_exec
// One wave per SIMD (4 waves per workgroup, 1 workgroup): 4,096 independent-chain v_fma_f64 (8 chains) with the lanes
// beyond `active` switched off by a branch.  If the f64 pipe skipped empty 16-lane quarter-waves, a transform split over
// more waves with fewer active lanes each would shorten the latency kernels' single-wave inverse transforms.
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

__global__ __launch_bounds__(256) void k(unsigned long long *out, double *sink, int active, int iters) {
  const int lane = threadIdx.x & 63;
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-9 * (lane + i);
  const double m = 1.0000001, c = 1e-12;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  if (lane < active) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 1.2345) sink[0] = s;
  if (lane == 0) atomicMax(&out[0], t1 - t0);
}

int main() {
  unsigned long long *out;
  double *sink;
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&sink, 64));
  const int iters = 256;
  for (int active : {64, 48, 32, 16, 8, 1}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(out, 0, 64));
      hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, sink, active, iters);
      CK(hipDeviceSynchronize());
    }
    unsigned long long h = 0;
    CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
    printf("active lanes %2d: %.2f s_memtime ticks per v_fma_f64 (one wave per SIMD, 8 independent chains)\n", active, (double)h / (iters * 64.0));
  }
  return 0;
}
