// ubench_mfma.hip -- what clock and rate the matrix cores SUSTAIN on v_mfma_i32_32x32x32_i8 for the operand
// patterns of the key switch (not product code).
//
//   hipcc --offload-arch=gfx950 -O2 -o profiles/exp/ubench_mfma profiles/exp/ubench_mfma.hip && ./profiles/exp/ubench_mfma
//
// Every SIMD runs WPS waves of back-to-back independent MFMAs (8 accumulators) for ~0.3 s; the kernel samples
// s_memtime / s_memrealtime, the host samples hwmon power.  mode: 0 random A x random B, 1 one-hot A (one byte
// of every dword = 1) x random B, 2 one-hot A x B with every fourth byte zero (the k = 0 rows), 3 zeros x zeros,
// 4 = mode 1 with 16 v_add_u32 between groups of 8 MFMAs (issue slots used beside the pipe).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

__device__ unsigned xs(unsigned &s) {
  s ^= s << 13;
  s ^= s >> 17;
  s ^= s << 5;
  return s;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_mfma(unsigned long long *clk, int iters, int *sink) {
  unsigned s = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x * 7919u;
  i32x4 A[4], B[4];
  for (int i = 0; i < 4; ++i)
    for (int c = 0; c < 4; ++c) {
      const unsigned r = xs(s);
      unsigned a = r, b = xs(s);
      if (MODE == 1 || MODE == 2 || MODE == 4) a = 1u << ((r & 3u) * 8u);
      if (MODE == 2) b &= 0xFFFFFF00u;
      if (MODE == 3) a = b = 0;
      A[i][c] = (int)a;
      B[i][c] = (int)b;
    }
  i32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0;
  int w[16];
  for (int i = 0; i < 16; ++i) w[i] = threadIdx.x + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[(i + u) & 3], B[(i + 2 * u) & 3], acc[i], 0, 0, 0);
      if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 15]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int z = 0;
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) z += acc[i][e];
  for (int i = 0; i < 16; ++i) z += w[i];
  if (z == 0x12345678) sink[0] = z;
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&clk[0], t1 - t0);
    atomicAdd(&clk[1], r1 - r0);
  }
}

static std::vector<std::string> power_files() {
  std::vector<std::string> v;
  glob_t g;
  if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", 0, nullptr, &g) == 0)
    for (size_t i = 0; i < g.gl_pathc; ++i) v.push_back(g.gl_pathv[i]);
  globfree(&g);
  return v;
}

template <int MODE>
void run(const char *name, int blocks, int iters, unsigned long long *d_clk, int *d_sink, int rtc_khz) {
  auto pf = power_files();
  std::atomic<bool> stop{false};
  double wmax = 0, wsum = 0;
  int wn = 0;
  std::thread th([&] {
    while (!stop.load()) {
      double best = 0;
      for (auto &f : pf) {
        std::ifstream in(f);
        double x = 0;
        if (in >> x) best = std::max(best, x * 1e-6);
      }
      if (best > 0) {
        wmax = std::max(wmax, best);
        wsum += best;
        ++wn;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
  });
  CK(hipMemset(d_clk, 0, 64));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_mfma<MODE>, dim3(blocks), dim3(256), 0, 0, d_clk, iters / 8, d_sink);  // warm-up
  CK(hipDeviceSynchronize());
  CK(hipMemset(d_clk, 0, 64));
  wmax = wsum = 0;
  wn = 0;
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k_mfma<MODE>, dim3(blocks), dim3(256), 0, 0, d_clk, iters, d_sink);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  stop = true;
  th.join();
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  unsigned long long h[2];
  CK(hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost));
  const double mhz = h[1] ? (double)h[0] / (double)h[1] * (rtc_khz / 1000.0) : 0;
  const double mfmas = (double)blocks * 4 * iters * 32.0;
  const double tops = mfmas * 2.0 * 32 * 32 * 32 / (ms * 1e-3) * 1e-12;
  // pipe utilisation at the sampled clock: 32 cycles per instruction per SIMD
  const double waves_per_simd = (double)blocks * 4 / 1024.0;
  const double util = mfmas / 1024.0 * 32.0 / (mhz * 1e6 * ms * 1e-3);
  printf("%-44s %8.2f ms  %7.1f TOPS  %6.0f MHz  util %.2f  power avg %5.0f max %5.0f W  (%.1f waves/SIMD)\n", name, ms, tops,
         mhz, util, wn ? wsum / wn : 0.0, wmax, waves_per_simd);
}

int main() {
  int rtc_khz = 100000;
  CK(hipDeviceGetAttribute(&rtc_khz, hipDeviceAttributeWallClockRate, 0));
  unsigned long long *d_clk;
  int *d_sink;
  CK(hipMalloc((void **)&d_clk, 64));
  CK(hipMalloc((void **)&d_sink, 64));
  const int iters = 60000;  // x 32 MFMAs x 32 cycles = 61 M cycles ~ 30 ms per wave... scaled below
  for (int blocks : {256, 512}) {
    run<0>("random A x random B", blocks, iters, d_clk, d_sink, rtc_khz);
    run<1>("one-hot A x random B", blocks, iters, d_clk, d_sink, rtc_khz);
    run<2>("one-hot A x B with zero k=0 bytes", blocks, iters, d_clk, d_sink, rtc_khz);
    run<3>("zeros", blocks, iters, d_clk, d_sink, rtc_khz);
    run<4>("one-hot A x random B + 16 VALU per 8 MFMA", blocks, iters, d_clk, d_sink, rtc_khz);
  }
  // short launches (~7 ms, the key switch's length) after 50 ms of idle: does the clock get to its sustained value?
  for (int rep = 0; rep < 4; ++rep) {
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    CK(hipMemset(d_clk, 0, 64));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_mfma<2>, dim3(512), dim3(256), 0, 0, d_clk, 5500, d_sink);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[2];
    CK(hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost));
    printf("short launch after idle: %.2f ms  %.0f MHz\n", ms, h[1] ? (double)h[0] / (double)h[1] * (rtc_khz / 1000.0) : 0.0);
  }
  // back to back: 6 short launches without a gap
  {
    hipEvent_t ev[7];
    for (auto &e : ev) CK(hipEventCreate(&e));
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    CK(hipEventRecord(ev[0]));
    for (int i = 0; i < 6; ++i) {
      hipLaunchKernelGGL(k_mfma<2>, dim3(512), dim3(256), 0, 0, d_clk, 5500, d_sink);
      CK(hipEventRecord(ev[i + 1]));
    }
    CK(hipDeviceSynchronize());
    for (int i = 0; i < 6; ++i) {
      float ms = 0;
      CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
      printf("back-to-back short launch %d: %.2f ms\n", i, ms);
    }
  }
  return 0;
}
