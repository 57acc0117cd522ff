// ubench_lds.hip -- LDS store / atomic cost for the latency kernel's partial-product exchange (not product code).
//   hipcc --offload-arch=gfx950 -O2 -o profiles/exp/ubench_lds profiles/exp/ubench_lds.hip && ./profiles/exp/ubench_lds
// W waves of one workgroup each issue 256 x 16 LDS operations; cycles per wave-instruction seen by wave 0.
//   mode 0: ds_write_b128 into wave-private regions     (the exchange of round 2: 16 per wave per step)
//   mode 1: ds_add_f64 (no return) of every wave into the SAME region (accumulate in place)
//   mode 2: ds_add_f64 into wave-private regions
//   mode 3: ds_read_b128
//   mode 4: ds_add_u32 into the same region
//   modes 5-11: narrower / split stores and loads (is the 16-byte store the expensive form?)
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
__global__ __launch_bounds__(512) void k(unsigned long long *out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double d[4] = {1.0 + lane, 2.0, 3.0, 4.0};
  const unsigned priv = (unsigned)(size_t)lds + wave * 8192;
  const unsigned shared = (unsigned)(size_t)lds;
  for (int i = threadIdx.x; i < 64 * 1024 / 8; i += blockDim.x) reinterpret_cast<double *>(lds)[i] = 0.0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const unsigned off = (unsigned)(u & 7) * 1024u;
      if (MODE == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(priv + off + lane * 16), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[0]))));
      if (MODE == 1) asm volatile("ds_add_f64 %0, %1" ::"v"(shared + off + lane * 16 + (u >> 3) * 8), "v"(d[0]));
      if (MODE == 2) asm volatile("ds_add_f64 %0, %1" ::"v"(priv + off + lane * 16 + (u >> 3) * 8), "v"(d[0]));
      if (MODE == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[2]))) : "v"(priv + off + lane * 16));
      if (MODE == 4) asm volatile("ds_add_u32 %0, %1" ::"v"(shared + off + lane * 4), "v"(lane));
      if (MODE == 5) asm volatile("ds_write_b64 %0, %1" ::"v"(priv + off + lane * 8), "v"(d[0]));
      if (MODE == 6) asm volatile("ds_write2_b64 %0, %1, %2 offset1:64" ::"v"(priv + off + lane * 8), "v"(d[0]), "v"(d[1]));
      if (MODE == 7) asm volatile("ds_write_b32 %0, %1" ::"v"(priv + off + lane * 4), "v"(lane));
      if (MODE == 8) asm volatile("ds_read_b64 %0, %1" : "=v"(d[2]) : "v"(priv + off + lane * 8));
      if (MODE == 9) asm volatile("ds_read2_b64 %0, %1 offset1:64" : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[2]))) : "v"(priv + off + lane * 8));
      if (MODE == 10) asm volatile("ds_write_b128 %0, %1" ::"v"(priv + off + ((lane & 31) * 32 + (lane >> 5) * 16)), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[0]))));
      if (MODE == 11) asm volatile("ds_write_b96 %0, %1" ::"v"(priv + off + lane * 16), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(3))) int *>(&d[0]))));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (d[2] == 1.2345) out[7] = 1;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
void run(const char *name, unsigned long long *d_out) {
  for (int waves : {1, 2, 4, 6, 8}) {
    const int iters = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, d_out, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, d_out, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    printf("  %-44s %d wave(s): %6.2f cycles per wave-instruction, %6.2f per instruction of the workgroup\n", name, waves,
           (double)h[0] / (iters * 16.0), (double)h[0] / (iters * 16.0 * waves));
  }
}

int main() {
  unsigned long long *d;
  CK(hipMalloc(&d, 64));
  run<0>("ds_write_b128, wave-private", d);
  run<1>("ds_add_f64, every wave into the same region", d);
  run<2>("ds_add_f64, wave-private", d);
  run<3>("ds_read_b128", d);
  run<4>("ds_add_u32, same region", d);
  run<5>("ds_write_b64", d);
  run<6>("ds_write2_b64 (2 x 8 B, 512 B apart)", d);
  run<7>("ds_write_b32", d);
  run<8>("ds_read_b64", d);
  run<9>("ds_read2_b64 (2 x 8 B, 512 B apart)", d);
  run<10>("ds_write_b128, lanes l / l+32 adjacent", d);
  run<11>("ds_write_b96", d);
  return 0;
}
