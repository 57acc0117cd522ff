// ubench.hip -- gfx950 micro-measurements behind the blind-rotation design decisions (not product code).
//
//   hipcc --offload-arch=gfx950 -O2 -o profiles/exp/ubench profiles/exp/ubench.hip && ./profiles/exp/ubench
//
// Part 1: issue cost (shader cycles per wave-instruction) of the candidate instructions, one wave alone and
//         two waves of one workgroup, from s_memtime around an unrolled independent stream.
// Part 2: the shader clock and board power the chip SUSTAINS when every SIMD runs a given instruction mix at
//         two waves per SIMD for about a second (s_memtime / s_memrealtime in the kernel; hwmon power1 on the
//         host side when readable).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

enum Op {
  FMA64, ADD64, MUL64, CVT64, ADD32, MOV32, BFE32, DPP_ROR8, DPP_QUAD, PERM32SWAP, PERM16SWAP, BPERMUTE, SWIZZLE,
  DSW128, DSR128, DSW64, DSR64, FMA32, PKFMA32, NOPS
};
static const char *kOpName[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_f64_i32", "v_add_u32", "v_mov_b32",
                                "v_bfe_i32", "v_mov_b32_dpp row_ror:8", "v_mov_b32_dpp quad_perm",
                                "v_permlane32_swap_b32", "v_permlane16_swap_b32", "ds_bpermute_b32", "ds_swizzle_b32",
                                "ds_write_b128", "ds_read_b128", "ds_write_b64", "ds_read_b64", "v_fma_f32",
                                "v_pk_fma_f32"};

// 16 independent instances of one instruction (register operands rotate so nothing depends on its predecessor)
template <int OP>
__device__ __forceinline__ void body(double (&d)[16], int (&w)[16], double c, unsigned lds_addr) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(c));
    if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
    if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
    if (OP == CVT64) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(w[i]));
    if (OP == ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 15]));
    if (OP == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "v"(w[(i + 5) & 15]));
    if (OP == BFE32) asm volatile("v_bfe_i32 %0, %0, 3, 6" : "+v"(w[i]));
    if (OP == DPP_ROR8) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(w[i]) : "v"(w[(i + 5) & 15]));
    if (OP == DPP_QUAD) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(w[i]) : "v"(w[(i + 5) & 15]));
    if (OP == PERM32SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(w[i]), "+v"(w[(i + 8) & 15]));
    if (OP == PERM16SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(w[i]), "+v"(w[(i + 8) & 15]));
    if (OP == BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(w[i]) : "v"(lds_addr));
    if (OP == SWIZZLE) asm volatile("ds_swizzle_b32 %0, %0 offset:0x041f" : "+v"(w[i]));
    if (OP == DSW128) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lds_addr), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[i & 14]))), "n"(0));
    if (OP == DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[i & 14]))) : "v"(lds_addr));
    if (OP == DSW64) asm volatile("ds_write_b64 %0, %1" ::"v"(lds_addr), "v"(d[i]));
    if (OP == DSR64) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(lds_addr));
    if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 15]));
    if (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(c));
  }
}

template <int OP>
__global__ __launch_bounds__(128) void k_issue(unsigned long long *out, int iters, double c) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  double d[16];
  int w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    d[i] = c * (i + 1) + threadIdx.x;
    w[i] = (int)threadIdx.x * 7 + i;
  }
  const unsigned lds_addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {  // 256 instructions per trip: the loop branch is noise
#pragma unroll
    for (int u = 0; u < 16; ++u) body<OP>(d, w, c, lds_addr);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  int z = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s += d[i];
    z += w[i];
  }
  if (s == 1.2345 && z == 77) out[7] = 1;  // keep the work alive
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

// dependent chain: latency of back-to-back dependent f64 FMAs
__global__ void k_chain(unsigned long long *out, int iters, double c) {
  double x = c + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(c));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (x == 1.2345) out[7] = 1;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

// Part 2: sustained mix.  mode: 0 fma64, 1 add64, 2 mul64, 3 int32, 4 fma64 + LDS b128 write/read, 5 half fma half add
template <int MODE>
__global__ __launch_bounds__(64, 2) void k_mix(unsigned long long *clk, int iters, double c) {
  __shared__ __attribute__((aligned(16))) char lds[16 * 1024];
  double d[16];
  int w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    d[i] = c * (i + 1) + threadIdx.x;
    w[i] = (int)threadIdx.x * 7 + i;
  }
  const unsigned lds_addr = (unsigned)(size_t)lds + threadIdx.x * 16;
  double rnd[8];  // random mantissas, magnitudes alternating around 1 so that products stay bounded
  {
    unsigned long long x = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1) + blockIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const unsigned long long mant = x & 0x000FFFFFFFFFFFFFull;
      // |r| in [0.5, 1): d = d * r + r' is a contraction, so the operands stay finite with random mantissas forever
      const unsigned long long bits = 0x3FE0000000000000ull | mant | ((i & 2) ? 0x8000000000000000ull : 0ull);
      rnd[i] = __longlong_as_double((long long)bits);
    }
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { body<FMA64>(d, w, c, lds_addr); body<FMA64>(d, w, c, lds_addr); }
    if (MODE == 1) { body<ADD64>(d, w, c, lds_addr); body<ADD64>(d, w, c, lds_addr); }
    if (MODE == 2) { body<MUL64>(d, w, c, lds_addr); body<MUL64>(d, w, c, lds_addr); }
    if (MODE == 3) { body<ADD32>(d, w, c, lds_addr); body<ADD32>(d, w, c, lds_addr); }
    if (MODE == 4) {
      body<FMA64>(d, w, c, lds_addr);
      body<FMA64>(d, w, c, lds_addr);
      asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[0]))));
      asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(lds_addr), "v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[2]))));
      asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[4]))) : "v"(lds_addr));
      asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(4))) int *>(&d[6]))) : "v"(lds_addr));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (MODE == 5) { body<FMA64>(d, w, c, lds_addr); body<ADD64>(d, w, c, lds_addr); }
    if (MODE == 6) {  // FMAs on operands with full random mantissas: d[i] = d[i] * r[j] + r[k], |r| < 1 (bounded)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(rnd[(i + u) & 7]), "v"(rnd[(i + 3 + u) & 7]));
    }
    if (MODE == 7) {  // adds on random mantissas: d[i] = r[j] - d[i] (bounded)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %1, -%0" : "+v"(d[i]) : "v"(rnd[(i + u) & 7]));
    }
    if (MODE == 9) {  // multiplies on random mantissas: d[i] = r[j] * r[k] (fresh every time)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[i]) : "v"(rnd[(i + u) & 7]), "v"(rnd[(i + 3 + u) & 7]));
    }
    if (MODE == 8) {  // FMAs whose multiplier operand is the constant 2.0 (the Y = 2a - X half of the butterflies)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, 2.0, %1, -%0" : "+v"(d[i]) : "v"(rnd[(i + u) & 7]));  // d = 2r - d (bounded)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  int z = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s += d[i];
    z += w[i];
  }
  if (s == 1.2345 && z == 77) clk[7] = 1;
  if (threadIdx.x == 0) {
    atomicAdd(&clk[0], t1 - t0);
    atomicAdd(&clk[1], r1 - r0);
  }
}

// ---- host ------------------------------------------------------------------------------------------------
static std::vector<std::string> power_files() {
  std::vector<std::string> v;
  for (const char *pat : {"/sys/class/drm/card*/device/hwmon/hwmon*/power1_average",
                          "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"}) {
    glob_t g;
    if (glob(pat, 0, nullptr, &g) == 0)
      for (size_t i = 0; i < g.gl_pathc; ++i) v.push_back(g.gl_pathv[i]);
    globfree(&g);
  }
  return v;
}
static double read_uW(const std::string &f) {
  std::ifstream in(f);
  double x = -1;
  in >> x;
  return x;
}

template <int OP>
static void issue(unsigned long long *d_out, int waves) {
  const int iters = 200;
  unsigned long long h[8];
  hipLaunchKernelGGL(k_issue<OP>, dim3(1), dim3(64 * waves), 0, 0, d_out, iters, 1.0000001);  // warm
  hipLaunchKernelGGL(k_issue<OP>, dim3(1), dim3(64 * waves), 0, 0, d_out, iters, 1.0000001);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
  printf("  %-28s %d wave(s): %6.2f shader cycles per wave-instruction\n", kOpName[OP], waves,
         (double)h[0] / (iters * 256.0));
}

template <int MODE>
static void mix(unsigned long long *d_clk, const char *name, int cus, int rtc_khz, const std::vector<std::string> &pf) {
  const int iters = 6000000;  // x 32 instructions x 4 cycles ~ 0.15 G cycles per wave; 2 waves per SIMD share the SIMD
  CK(hipMemset(d_clk, 0, 64));
  std::atomic<bool> stop{false};
  std::vector<double> samples;
  std::thread th([&] {
    while (!stop.load()) {
      double best = 0;  // the busiest GPU of the box is ours
      for (auto &f : pf) {
        double x = read_uW(f);
        if (x * 1e-6 > best) best = x * 1e-6;
      }
      samples.push_back(best);
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  });
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k_mix<MODE>, dim3(cus * 8), dim3(64), 0, 0, d_clk, iters, 1.0000001);
  CK(hipDeviceSynchronize());
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  stop = true;
  th.join();
  unsigned long long h[2];
  CK(hipMemcpy(h, d_clk, sizeof(h), hipMemcpyDeviceToHost));
  double pw = 0, pmax = 0;
  // skip the first quarter of the samples (ramp)
  size_t n0 = samples.size() / 4, n = 0;
  for (size_t i = n0; i < samples.size(); ++i) {
    pw += samples[i];
    pmax = samples[i] > pmax ? samples[i] : pmax;
    ++n;
  }
  printf("  %-34s %8.1f ms  shader clock %7.1f MHz  issue %5.1f%%  power avg %6.0f W max %6.0f W (%zu samples)\n", name,
         ms, h[1] ? (double)h[0] / (double)h[1] * rtc_khz / 1000.0 : 0.0,
         h[0] ? 100.0 * (double)iters * 32 * 4 * 2 / ((double)h[0] / (cus * 8)) : 0.0, n ? pw / n : 0.0, pmax, n);
}

int main(int argc, char **argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int rtc_khz = 100000;
  (void)hipDeviceGetAttribute(&rtc_khz, hipDeviceAttributeWallClockRate, 0);
  printf("%s  CUs %d  clockRate %d kHz  wall clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount,
         prop.clockRate, rtc_khz);
  unsigned long long *d;
  CK(hipMalloc(&d, 64));
  printf("part 1: issue cost\n");
  for (int waves = 1; waves <= 2; ++waves) {
    issue<FMA64>(d, waves);
    issue<ADD64>(d, waves);
    issue<MUL64>(d, waves);
    issue<CVT64>(d, waves);
    issue<ADD32>(d, waves);
    issue<MOV32>(d, waves);
    issue<BFE32>(d, waves);
    issue<FMA32>(d, waves);
    issue<PKFMA32>(d, waves);
    issue<DPP_ROR8>(d, waves);
    issue<DPP_QUAD>(d, waves);
    issue<PERM32SWAP>(d, waves);
    issue<PERM16SWAP>(d, waves);
    issue<BPERMUTE>(d, waves);
    issue<SWIZZLE>(d, waves);
    issue<DSW128>(d, waves);
    issue<DSR128>(d, waves);
    issue<DSW64>(d, waves);
    issue<DSR64>(d, waves);
  }
  {
    unsigned long long h[8];
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, 2000, 1.0000001);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, 2000, 1.0000001);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("  dependent v_fma_f64 chain: %.2f cycles per instruction\n", (double)h[0] / (2000 * 16.0));
  }
  if (argc > 1 && !strcmp(argv[1], "issue")) return 0;
  auto pf = power_files();
  printf("part 2: sustained mix, %d CUs x 8 waves (2 per SIMD); power files: %zu\n", prop.multiProcessorCount, pf.size());
  for (int rep = 0; rep < 2; ++rep) {
    mix<0>(d, "all v_fma_f64", prop.multiProcessorCount, rtc_khz, pf);
    mix<1>(d, "all v_add_f64", prop.multiProcessorCount, rtc_khz, pf);
    mix<2>(d, "all v_mul_f64", prop.multiProcessorCount, rtc_khz, pf);
    mix<5>(d, "half v_fma_f64 half v_add_f64", prop.multiProcessorCount, rtc_khz, pf);
    mix<3>(d, "all v_add_u32", prop.multiProcessorCount, rtc_khz, pf);
    mix<4>(d, "v_fma_f64 + 4 LDS b128 per 32", prop.multiProcessorCount, rtc_khz, pf);
    mix<6>(d, "v_fma_f64, random mantissas", prop.multiProcessorCount, rtc_khz, pf);
    mix<7>(d, "v_add_f64, random mantissas", prop.multiProcessorCount, rtc_khz, pf);
    mix<8>(d, "v_fma_f64 with multiplier 2.0", prop.multiProcessorCount, rtc_khz, pf);
    mix<9>(d, "v_mul_f64, random mantissas", prop.multiProcessorCount, rtc_khz, pf);
  }
  return 0;
}
