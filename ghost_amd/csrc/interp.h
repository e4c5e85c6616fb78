// interp.h -- host-side design of the polyphase interpolators of the interpolating synthesis
// (synthi.hip).  No device code.
//
// A scale whose band is narrow against its level's rate is heavily oversampled at the full
// rate: its complex output is made at q x the level's rate (q phases of the 256-point inverse
// FFT), demodulated to its band centre, and brought to the full rate by a T-tap FIR with real
// coefficients,
//     y[I m' + rho] = sum_{j<T} c_rho[j] z[m' + j - (T/2 - 1)],      I = R / q,  rho in [0, I).
// |y| does not see the demodulation, so amplitude and power need nothing else.
// The coefficients are the least-squares fit, over the band |theta| <= alpha pi of the
// oversampled signal, of sum_j c_j e^{i theta (j - T/2 + 1)} to e^{i theta tau}: the normal
// equations are the T x T prolate matrix alpha sinc(alpha (j - k)) against alpha sinc(alpha
// (j - T/2 + 1 - tau)).  The matrix is ill-conditioned by nature (its small eigenvalues belong to
// sequences with no energy in the band, which is why their coefficients do not matter); a ridge
// of 1e-13 of the diagonal keeps the elimination well-defined in fp64.
#pragma once
#include <cmath>
#include <vector>

namespace gcwt {

constexpr int kInterpTaps = 8;       // T
constexpr int kInterpMaxFactor = 1024;  // I = R / q at most: a lane's coefficient set is (4 (wave-task mod 4) .. + lane) & (I/4 - 1),
                                        // and a wave's wave-tasks stay in one class mod 4 (synthi.hip)
#ifndef GCWT_SYNTHI_COLS
#define GCWT_SYNTHI_COLS 16             // (8 and 32 are measurement builds: profiles/r03_synth_study.md)
#endif
constexpr int kInterpCols = GCWT_SYNTHI_COLS;   // columns (block, scale slot, phase) of one pass of k_synthi (synthi.hip)
constexpr int kInterpMaxPhases = kInterpCols;   // q at most

inline double interp_sinc(double x) {   // sin(pi x) / (pi x)
  if (std::fabs(x) < 1e-12) return 1.0;
  const double a = M_PI * x;
  return std::sin(a) / a;
}

// c[rho * T + j] for tau = (rho - shift) / I (shift = 0.5: the half-sample delay that kernels
// of even length carry).
inline void design_interp(int T, int I, double alpha, double shift, double* c) {
  std::vector<double> A((size_t)T * T), M((size_t)T * (T + 1));
  for (int j = 0; j < T; ++j)
    for (int k = 0; k < T; ++k) A[(size_t)j * T + k] = alpha * interp_sinc(alpha * (double)(j - k));
  const double ridge = 1e-13 * alpha;
  for (int rho = 0; rho < I; ++rho) {
    const double tau = ((double)rho - shift) / (double)I;
    for (int j = 0; j < T; ++j) {
      for (int k = 0; k < T; ++k) M[(size_t)j * (T + 1) + k] = A[(size_t)j * T + k] + (j == k ? ridge : 0.0);
      M[(size_t)j * (T + 1) + T] = alpha * interp_sinc(alpha * ((double)(j - (T / 2 - 1)) - tau));
    }
    // Gaussian elimination with partial pivoting
    for (int col = 0; col < T; ++col) {
      int piv = col;
      for (int r = col + 1; r < T; ++r)
        if (std::fabs(M[(size_t)r * (T + 1) + col]) > std::fabs(M[(size_t)piv * (T + 1) + col])) piv = r;
      if (piv != col)
        for (int k = 0; k <= T; ++k) std::swap(M[(size_t)piv * (T + 1) + k], M[(size_t)col * (T + 1) + k]);
      const double d = M[(size_t)col * (T + 1) + col];
      for (int r = col + 1; r < T; ++r) {
        const double f = M[(size_t)r * (T + 1) + col] / d;
        if (f == 0.0) continue;
        for (int k = col; k <= T; ++k) M[(size_t)r * (T + 1) + k] -= f * M[(size_t)col * (T + 1) + k];
      }
    }
    for (int r = T - 1; r >= 0; --r) {
      double s = M[(size_t)r * (T + 1) + T];
      for (int k = r + 1; k < T; ++k) s -= M[(size_t)r * (T + 1) + k] * c[(size_t)rho * T + k];
      c[(size_t)rho * T + r] = s / M[(size_t)r * (T + 1) + r];
    }
  }
}

// |sum_j c_j e^{i theta (j - T/2 + 1)} - e^{i theta tau}| : what the interpolator does to a
// component at theta (radians per sample of the oversampled signal) away from the demodulation
// centre.
inline double interp_error_at(int T, const double* c, double tau, double theta) {
  double re = -std::cos(theta * tau), im = -std::sin(theta * tau);
  for (int j = 0; j < T; ++j) {
    const double ph = theta * (double)(j - (T / 2 - 1));
    re += c[j] * std::cos(ph);
    im += c[j] * std::sin(ph);
  }
  return std::sqrt(re * re + im * im);
}

}  // namespace gcwt
