// ubench_lds2.hip -- which waves of an eight-wave workgroup share the LDS store path?  (not product code)
//   hipcc --offload-arch=gfx950 -O2 -o profiles/exp/ubench_lds2 profiles/exp/ubench_lds2.hip && ./profiles/exp/ubench_lds2
// A workgroup of 8 waves; the waves in `mask` each issue 256 x 16 ds_write_b128 (or ds_read_b128) into private regions,
// the others wait at the final barrier.  Reported: cycles per wave-instruction of the SLOWEST active wave and the
// aggregate bytes per cycle; plus the hardware SIMD id of every wave (HW_REG_HW_ID bits 5:4).
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

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long *out, int iters, unsigned mask) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double d[4] = {1.0 + lane, 2.0, 3.0, 4.0};
  const unsigned priv = (unsigned)(size_t)lds + wave * 8192;
  for (int i = threadIdx.x; i < 64 * 1024 / 8; i += blockDim.x) reinterpret_cast<double *>(lds)[i] = 0.0;
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (lane == 0) out[16 + wave] = (hwid >> 4) & 3;
  __syncthreads();
  if ((mask >> wave) & 1) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const unsigned off = (unsigned)(u & 7) * 1024u;
        if (MODE == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(priv + off + lane * 16), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[0]))));
        if (MODE == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[2]))) : "v"(priv + off + lane * 16));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (d[2] == 1.2345) out[7] = 1;
    if (lane == 0) atomicMax(&out[0], t1 - t0);
  }
  __syncthreads();
}

template <int MODE>
void run(const char *name, unsigned long long *d_out) {
  const unsigned masks[] = {0x01, 0x03, 0x05, 0x11, 0x0F, 0x33, 0x55, 0x3F, 0xFC, 0xFF};
  for (unsigned mask : masks) {
    const int iters = 256;
    CK(hipMemset(d_out, 0, 256));
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(512), 0, 0, d_out, iters, mask);
    CK(hipMemset(d_out, 0, 8));
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(512), 0, 0, d_out, iters, mask);
    CK(hipDeviceSynchronize());
    unsigned long long h[32];
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    const int nw = __builtin_popcount(mask);
    printf("  %-14s waves 0x%02X (%d): %6.2f cycles per wave-instruction (slowest wave), %6.1f B/cycle aggregate   simd of waves 0..7:",
           name, mask, nw, (double)h[0] / (iters * 16.0), nw * 1024.0 * iters * 16.0 / (double)h[0]);
    for (int w = 0; w < 8; ++w) printf(" %llu", h[16 + w]);
    printf("\n");
  }
}

int main() {
  unsigned long long *d;
  CK(hipMalloc(&d, 256));
  run<0>("ds_write_b128", d);
  run<1>("ds_read_b128", d);
  return 0;
}
