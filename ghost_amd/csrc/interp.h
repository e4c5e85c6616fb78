// interp.h -- host-side design of the polyphase interpolators of the interpolating synthesis
// (synthi.hip).  No device code.
//
// A scale whose band is narrow against its level's rate is heavily oversampled at the full
// rate: its complex output is made at q x the level's rate (q phases of the 256-point inverse
// FFT), demodulated to its band centre, and brought to the full rate by a T-tap FIR with real
// coefficients,
//     y[I m' + rho] = sum_{j<T} c_rho[j] z[m' + j - (T/2 - 1)],      I = R / q,  rho in [0, I).
// |y| does not see the demodulation, so amplitude and power need nothing else.
// The coefficients are a weighted least-squares fit of sum_j c_j e^{i theta (j - T/2 + 1)} to
// e^{i theta tau}, weighted by the level's own gains (design_interp_weighted below).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace gcwt {

constexpr int kInterpTaps = 8;       // T at most, and the width of a coefficient row (six taps sit in its middle)
constexpr int kInterpMaxFactor = 1024;  // I = R / q at most: a lane's coefficient set is (4 (wave-task mod 4) .. + lane) & (I/4 - 1),
                                        // and a wave's wave-tasks stay in one class mod 4 (synthi.hip)
#ifndef GCWT_SYNTHI_COLS
#define GCWT_SYNTHI_COLS 16             // (8 and 32 are measurement builds: profiles/r03_synth_study.md)
#endif
constexpr int kInterpCols = GCWT_SYNTHI_COLS;   // columns (block, scale slot, phase) of one pass of k_synthi (synthi.hip)
constexpr int kInterpMaxPhases = kInterpCols;   // q at most

inline double interp_error_at_fwd(int T, const double* c, double tau, double theta);

// Least squares  min |A x - b|  for a tall m x n matrix (row-major), by Householder reflections:
// factor once (the interpolator design below keeps its weights fixed over all sub-sample
// positions), then one solve per right-hand side.
struct TallQR {
  int m = 0, n = 0;
  std::vector<double> a;      // the reflectors below the diagonal, R on and above it
  std::vector<double> beta;
  void factor(const std::vector<double>& mat, int rows, int cols) {
    m = rows; n = cols; a = mat; beta.assign((size_t)n, 0.0);
    for (int k = 0; k < n; ++k) {
      double norm = 0.0;
      for (int i = k; i < m; ++i) norm += a[(size_t)i * n + k] * a[(size_t)i * n + k];
      norm = std::sqrt(norm);
      if (norm == 0.0) continue;
      const double alpha = a[(size_t)k * n + k] > 0 ? -norm : norm;
      const double v0 = a[(size_t)k * n + k] - alpha;
      a[(size_t)k * n + k] = alpha;
      // v = (v0, a[k+1..m, k]); beta = 2 / |v|^2; stored with v0 kept aside as beta's partner
      double vv = v0 * v0;
      for (int i = k + 1; i < m; ++i) vv += a[(size_t)i * n + k] * a[(size_t)i * n + k];
      beta[(size_t)k] = vv > 0 ? 2.0 / vv : 0.0;
      v0s.resize((size_t)n);
      v0s[(size_t)k] = v0;
      for (int j = k + 1; j < n; ++j) {
        double dot = v0 * a[(size_t)k * n + j];
        for (int i = k + 1; i < m; ++i) dot += a[(size_t)i * n + k] * a[(size_t)i * n + j];
        dot *= beta[(size_t)k];
        a[(size_t)k * n + j] -= dot * v0;
        for (int i = k + 1; i < m; ++i) a[(size_t)i * n + j] -= dot * a[(size_t)i * n + k];
      }
    }
  }
  void solve(std::vector<double>& rhs, double* x) const {     // rhs (m) is overwritten
    for (int k = 0; k < n; ++k) {
      if (beta[(size_t)k] == 0.0) continue;
      double dot = v0s[(size_t)k] * rhs[(size_t)k];
      for (int i = k + 1; i < m; ++i) dot += a[(size_t)i * n + k] * rhs[(size_t)i];
      dot *= beta[(size_t)k];
      rhs[(size_t)k] -= dot * v0s[(size_t)k];
      for (int i = k + 1; i < m; ++i) rhs[(size_t)i] -= dot * a[(size_t)i * n + k];
    }
    for (int r = n - 1; r >= 0; --r) {
      double sacc = rhs[(size_t)r];
      for (int k = r + 1; k < n; ++k) sacc -= a[(size_t)r * n + k] * x[k];
      const double d = a[(size_t)r * n + r];
      x[r] = d != 0.0 ? sacc / d : 0.0;
    }
  }
  std::vector<double> v0s;
};

// The T-tap interpolators of a level, c[rho * T + j] for tau = (rho - shift) / I (shift = 0.5: the
// half-sample delay that kernels of even length carry), fitted where it matters:
// weighted least squares of sum_j c_j e^{i theta (j - T/2 + 1)} against e^{i theta tau} over the
// bins d of the level's grid (theta = 2 pi d / (B q), d counted from a scale's demodulation
// centre), weighted by `genv[d + B]`, the largest gain any scale of the level has at that
// distance -- times the weights Lawson's iteration arrives at for the hardest delay, tau = 1/2,
// which turn the fit into a minimax one on gain x error.  The weights are shared by every tau, so
// the matrix is factored once (TallQR) and each sub-sample position is one solve.  Against the
// unweighted fit on a band this buys two taps or a factor two in oversampling:
//   q = 4: 6 taps 5e-8 .. 7e-8 (8 taps unweighted: 4e-8);  q = 2: 8 taps 9e-8 .. 1.3e-7.
inline void design_interp_weighted(int T, int I, int q, int B, const double* genv, double shift, double* c) {
  // every second bin: the envelope is smooth on that scale and the demodulation centres are even
  std::vector<double> th, ge;
  for (int d = -B; d < B; d += 2) {
    const double g = std::max(genv[d + B], d + 1 < B ? genv[d + 1 + B] : 0.0);
    if (!(g > 1e-12)) continue;
    th.push_back(std::remainder(2.0 * M_PI * (double)d / ((double)B * q), 2.0 * M_PI));
    ge.push_back(g);
  }
  const int n = (int)th.size();
  if (n == 0) { for (int i = 0; i < I * T; ++i) c[i] = 0.0; return; }
  std::vector<double> w = ge, mat((size_t)2 * n * T), rhs((size_t)2 * n), x((size_t)T);
  std::vector<double> cs((size_t)n * T), sn((size_t)n * T);   // e^{i theta (j - T/2 + 1)}: the same in every pass
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < T; ++j) {
      const double ph = th[(size_t)i] * (double)(j - (T / 2 - 1));
      cs[(size_t)i * T + j] = std::cos(ph);
      sn[(size_t)i * T + j] = std::sin(ph);
    }
  TallQR qr;
  auto build = [&]() {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < T; ++j) {
        mat[(size_t)i * T + j] = w[(size_t)i] * cs[(size_t)i * T + j];
        mat[(size_t)(n + i) * T + j] = w[(size_t)i] * sn[(size_t)i * T + j];
      }
    qr.factor(mat, 2 * n, T);
  };
  std::vector<double> half_c((size_t)n), half_s((size_t)n);
  for (int i = 0; i < n; ++i) { half_c[(size_t)i] = std::cos(0.5 * th[(size_t)i]); half_s[(size_t)i] = std::sin(0.5 * th[(size_t)i]); }
  for (int it = 0; it < 16; ++it) {            // Lawson: weights towards the minimax fit at tau = 1/2
    build();
    for (int i = 0; i < n; ++i) { rhs[(size_t)i] = w[(size_t)i] * half_c[(size_t)i]; rhs[(size_t)(n + i)] = w[(size_t)i] * half_s[(size_t)i]; }
    qr.solve(rhs, x.data());
    double emax = 0.0;
    std::vector<double> e((size_t)n);
    for (int i = 0; i < n; ++i) {
      double re = -half_c[(size_t)i], im = -half_s[(size_t)i];
      for (int j = 0; j < T; ++j) { re += x[(size_t)j] * cs[(size_t)i * T + j]; im += x[(size_t)j] * sn[(size_t)i * T + j]; }
      e[(size_t)i] = ge[(size_t)i] * std::sqrt(re * re + im * im);
      emax = std::max(emax, e[(size_t)i]);
    }
    if (!(emax > 0.0)) break;
    double wmax = 0.0;
    for (int i = 0; i < n; ++i) {
      w[(size_t)i] *= std::sqrt(1e-30 + e[(size_t)i] / emax);
      wmax = std::max(wmax, w[(size_t)i]);
    }
    for (int i = 0; i < n; ++i) w[(size_t)i] /= wmax;
  }
  build();
  // e^{i theta tau} for tau = (rho - shift) / I by rotation from one sub-sample position to the next
  std::vector<double> pc((size_t)n), ps((size_t)n), dc((size_t)n), ds((size_t)n);
  for (int i = 0; i < n; ++i) {
    pc[(size_t)i] = std::cos(-th[(size_t)i] * shift / (double)I);
    ps[(size_t)i] = std::sin(-th[(size_t)i] * shift / (double)I);
    dc[(size_t)i] = std::cos(th[(size_t)i] / (double)I);
    ds[(size_t)i] = std::sin(th[(size_t)i] / (double)I);
  }
  for (int rho = 0; rho < I; ++rho) {
    for (int i = 0; i < n; ++i) {
      rhs[(size_t)i] = w[(size_t)i] * pc[(size_t)i];
      rhs[(size_t)(n + i)] = w[(size_t)i] * ps[(size_t)i];
      const double nc = pc[(size_t)i] * dc[(size_t)i] - ps[(size_t)i] * ds[(size_t)i];
      ps[(size_t)i] = pc[(size_t)i] * ds[(size_t)i] + ps[(size_t)i] * dc[(size_t)i];
      pc[(size_t)i] = nc;
    }
    qr.solve(rhs, c + (size_t)rho * T);
  }
}

// |sum_j c_j e^{i theta (j - T/2 + 1)} - e^{i theta tau}| : what the interpolator does to a
// component at theta (radians per sample of the oversampled signal) away from the demodulation
// centre.
inline double interp_error_at(int T, const double* c, double tau, double theta) {
  return interp_error_at_fwd(T, c, tau, theta);
}
inline double interp_error_at_fwd(int T, const double* c, double tau, double theta) {
  double re = -std::cos(theta * tau), im = -std::sin(theta * tau);
  for (int j = 0; j < T; ++j) {
    const double ph = theta * (double)(j - (T / 2 - 1));
    re += c[j] * std::cos(ph);
    im += c[j] * std::sin(ph);
  }
  return std::sqrt(re * re + im * im);
}

}  // namespace gcwt
