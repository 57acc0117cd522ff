// CPU check of the device FFT's algebra: the pass / transpose functions of
// rs-tfhe_amd/csrc/fft512.hpp are compiled for the host (empty qualifiers) and driven in
// lock-step over 64 emulated lanes with the LDS tile as a plain array, using the very twiddle table
// the kernels read (twiddles_host.hpp).  Checked against a direct long-double DFT of the twisted
// fold, i.e. the definition of KlemsaProcessor::ifft / fft (src/fft/klemsa.rs:88-150), in the
// engine's bin order bin_of(lane, slot).
//
// This is test infrastructure; it cannot replace the GPU parity tests (no device instruction runs
// here), it only keeps the index maps, constants and butterfly networks honest without a GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct double2 {
  double x, y;
};
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
#define __device__
#define __host__
#define __forceinline__ inline
#define __syncthreads()
#define TFHE_FFT_HOST_EMU 1
#include "../../rs-tfhe_amd/csrc/fft512.hpp"
#include "../../rs-tfhe_amd/csrc/twiddles_host.hpp"

using namespace tfhe;

struct Lane {
  double re[8], im[8];
};

int main() {
  std::vector<double2> table;
  make_twiddles(table);
  if ((int)table.size() != kTwEntries) {
    printf("table size %zu != %d\n", table.size(), kTwEntries);
    return 1;
  }
  std::vector<double2> t2(64), tile(kTileCplx);
  Twiddles tw[64];
  for (int l = 0; l < 64; ++l) tw[l].load(table.data(), t2.data(), l);

  const long double pi = 3.14159265358979323846264338327950288L;
  srand(7);
  int fails = 0;
  for (int trial = 0; trial < 4; ++trial) {
    // digits in [-32, 32) for the bootstrap case, full-range i32 for the stage API
    std::vector<long double> zr(512), zi(512);
    for (int j = 0; j < 512; ++j) {
      if (trial < 2) {
        zr[j] = (rand() % 64) - 32;
        zi[j] = (rand() % 64) - 32;
      } else {
        zr[j] = (long double)((int32_t)((uint32_t)rand() * 2654435761u));
        zi[j] = (long double)((int32_t)((uint32_t)rand() * 40503u * 65537u));
      }
    }
    // reference: Z[k] = sum_j z[j] * exp(i*pi*j/1024) * exp(-2*pi*i*j*k/512)
    std::vector<long double> Zr(512), Zi(512);
    long double zmax = 0;
    for (int k = 0; k < 512; ++k) {
      long double sr = 0, si = 0;
      for (int j = 0; j < 512; ++j) {
        const long e = ((long)j * (1 - 4 * k)) % 2048;
        const long double ang = pi * (long double)e / 1024.0L, c = cosl(ang), s = sinl(ang);
        sr += zr[j] * c - zi[j] * s;
        si += zr[j] * s + zi[j] * c;
      }
      Zr[k] = sr;
      Zi[k] = si;
      zmax = fmaxl(zmax, fmaxl(fabsl(sr), fabsl(si)));
    }
    // device algorithm, lock-step
    Lane L[64];
    for (int l = 0; l < 64; ++l)
      for (int m = 0; m < 8; ++m) {
        L[l].re[m] = (double)zr[l + 64 * m];
        L[l].im[m] = (double)zi[l + 64 * m];
      }
    for (int l = 0; l < 64; ++l) { fwd_pass1(L[l].re, L[l].im); tpA_write(L[l].re, L[l].im, tile.data(), l); }
    for (int l = 0; l < 64; ++l) { tpA_read(L[l].re, L[l].im, tile.data(), l); fwd_pass2(L[l].re, L[l].im, tw[l], l); }
    for (int l = 0; l < 64; ++l) tpB_write(L[l].re, L[l].im, tile.data(), l);
    for (int l = 0; l < 64; ++l) { tpB_read(L[l].re, L[l].im, tile.data(), l); fwd_pass3(L[l].re, L[l].im, tw[l]); }
    long double ferr = 0;
    for (int l = 0; l < 64; ++l)
      for (int s = 0; s < 8; ++s) {
        const int k = bin_of(l, s);
        ferr = fmaxl(ferr, fmaxl(fabsl(L[l].re[s] - Zr[k]), fabsl(L[l].im[s] - Zi[k])));
      }
    // inverse of that spectrum: expect 512 * z back (un-normalised)
    for (int l = 0; l < 64; ++l) { dft8<true>(L[l].re, L[l].im); tpBi_write(L[l].re, L[l].im, tile.data(), l); }
    for (int l = 0; l < 64; ++l) { tpBi_read(L[l].re, L[l].im, tile.data(), l); inv_pass2(L[l].re, L[l].im, tw[l], l); }
    for (int l = 0; l < 64; ++l) tpAi_write(L[l].re, L[l].im, tile.data(), l);
    for (int l = 0; l < 64; ++l) { tpAi_read(L[l].re, L[l].im, tile.data(), l); inv_pass3(L[l].re, L[l].im, tw[l]); }
    long double ierr = 0, zin = 0;
    for (int l = 0; l < 64; ++l)
      for (int m = 0; m < 8; ++m) {
        ierr = fmaxl(ierr, fabsl(L[l].re[m] / 512.0 - zr[l + 64 * m]));
        ierr = fmaxl(ierr, fabsl(L[l].im[m] / 512.0 - zi[l + 64 * m]));
        zin = fmaxl(zin, fmaxl(fabsl(zr[l + 64 * m]), fabsl(zi[l + 64 * m])));
      }
    const double frel = (double)(ferr / zmax), irel = (double)(ierr / zin);
    printf("trial %d: forward max err %.3Le (rel %.2e)  round trip max err %.3Le (rel %.2e)\n", trial, ferr, frel, ierr,
           irel);
    if (frel > 1e-14 || irel > 1e-14) ++fails;
  }
  // tie rounding of the stage API (klemsa.rs:145-146: f64::round, half away from zero, as i64 as u32)
  struct { double x; uint32_t want; } ties[] = {
      {0.5, 1u}, {-0.5, 0xFFFFFFFFu}, {1.5, 2u}, {2.5, 3u}, {-2.5, 0xFFFFFFFDu}, {0.49999999999999994, 0u},
      {4294967295.5, 0u}, {4294967296.5, 1u}, {-4294967296.5, 0xFFFFFFFFu}, {2147483647.5, 0x80000000u},
      {-2147483648.5, 0x7FFFFFFFu}, {1e15 + 0.5, (uint32_t)((long long)(1e15 + 1.0))}, {3.0, 3u}, {-7.25, (uint32_t)-7}};
  for (auto &t : ties) {
    const uint32_t got = round_half_away_to_torus(t.x);
    const uint32_t want = (uint32_t)(long long)std::round(t.x);
    if (got != t.want || want != t.want) {
      printf("tie %.17g: got %08x want %08x (std::round %08x)\n", t.x, got, t.want, want);
      ++fails;
    }
  }
  if (fails) {
    printf("FAILED (%d)\n", fails);
    return 1;
  }
  printf("fft host emulation: all checks passed\n");
  return 0;
}
