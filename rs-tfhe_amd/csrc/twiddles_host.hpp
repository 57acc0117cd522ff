// twiddles_host.hpp -- host-side construction of the context's twiddle table (layout: fft512.hpp,
// struct Twiddles).  Plain C++ (no device code) so that the CPU test of the FFT algebra
// (tests/cpp/test_fft_host.cpp) builds the very table the kernels read.
#pragma once
#include <cmath>
#include <vector>

namespace tfhe {

inline void make_twiddles(std::vector<double2> &tw) {
  // layout: fft512.hpp (Twiddles).  Every entry is exp(i*pi*e/1024) for an integer e, evaluated in
  // long double from the reduced exponent.
  tw.resize(640);  // kTwEntries
  const long double pi = 3.14159265358979323846264338327950288L;
  auto w = [&](long e) {
    e %= 2048;
    if (e < 0) e += 2048;
    const long double ang = pi * (long double)e / 1024.0L;
    return make_double2((double)cosl(ang), (double)sinl(ang));
  };
  const long W8 = -256;  // exp(-2*pi*i/8) = exp(i*pi*(-256)/1024)
  for (int lane = 0; lane < 64; ++lane) {
    // forward pass 3: c = exp(i*pi*(1 - 4*kappa)/1024), kappa = k1 + 8*k2 for lane = k1*8 + k2
    const long c = 1 - 4 * ((lane >> 3) + 8 * (lane & 7));
    tw[0 * 64 + lane] = w(4 * c);
    tw[1 * 64 + lane] = w(2 * c);
    tw[2 * 64 + lane] = w(c);
    tw[3 * 64 + lane] = w(c + W8);
    // inverse pass 3: c = exp(2*pi*i*lane/512) = exp(i*pi*4*lane/1024), G = exp(-i*pi*lane/1024)
    const long ci = 4 * lane, g = -lane;
    tw[256 + 0 * 64 + lane] = w(g);
    tw[256 + 1 * 64 + lane] = w(g + 4 * ci);
    tw[256 + 2 * 64 + lane] = w(2 * ci);
    tw[256 + 3 * 64 + lane] = w(ci);
    tw[256 + 4 * 64 + lane] = w(ci - W8);
  }
  for (int k = 0; k < 8; ++k) {
    // forward pass 2: c = exp(i*pi*(1 - 4*k1)/128) = exp(i*pi*8*(1 - 4*k1)/1024)
    const long c = 8 * (1 - 4 * k);
    tw[576 + 0 * 8 + k] = w(4 * c);
    tw[576 + 1 * 8 + k] = w(2 * c);
    tw[576 + 2 * 8 + k] = w(c);
    tw[576 + 3 * 8 + k] = w(c + W8);
    // inverse pass 2: c = exp(2*pi*i*l1/64) = exp(i*pi*32*l1/1024)
    const long ci = 32 * k;
    tw[576 + 32 + 0 * 8 + k] = w(4 * ci);
    tw[576 + 32 + 1 * 8 + k] = w(2 * ci);
    tw[576 + 32 + 2 * 8 + k] = w(ci);
    tw[576 + 32 + 3 * 8 + k] = w(ci - W8);
  }
}

}  // namespace tfhe
