// morse_exact.h -- the frequency response of the kernel the reference convolves with,
// exactly (host and device, fp64).
//
// The reference samples the Morse spectrum on an L-point grid, keeps bins
// 0 .. round(L/2)-1, centres it and takes an L-point inverse DFT
// (ghost/wave/morseutils.py:117-149, :175-196); the resulting L taps psi[0..L) are used
// as an FIR filter in 'same' mode (ghost/sigtools/convolution.py:68-87).  The response of
// that FIR filter -- not the continuous Morse spectrum -- is what every output is made
// of.  With theta_j = 2 pi j / L and A_j the kept spectrum samples,
//
//   psi[n]   = (1/L) sum_j A_j exp(i theta_j (n - (L-1)/2))
//   H(theta) = sum_n psi[n] exp(-i theta (n - (L-1)//2))
//            = exp(-i theta d) * G(theta),      d = (L-1)/2 - (L-1)//2  (0 or 1/2)
//   G(theta) = (1/L) sum_j A_j * sin(L (theta_j - theta)/2) / sin((theta_j - theta)/2)
//
// G is REAL (it changes sign in the side lobes that truncating to L taps creates) and
// 2 pi-periodic for odd L, 2 pi-antiperiodic for even L (which the half-sample phase
// undoes).  For (gamma, beta) = (3, 20) G equals the continuous spectrum to 6e-9 of its
// peak; for wavelets with heavier tails it does not, and only G matches the reference.
//
// All arguments are rational multiples of pi and reduced in integers, so the values are
// good to a few ulp whatever L and the grid are.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace gcwt {

// sin(pi * num / den), den > 0, |num| * 2 and den * 2 within int64
__host__ __device__ inline double sinpi_ratio(int64_t num, int64_t den) {
  int64_t m = num % (2 * den);
  if (m < 0) m += 2 * den;                 // [0, 2 den)
  double sign = 1.0;
  if (m >= den) { m -= den; sign = -1.0; } // sin(pi + x) = -sin x
  if (2 * m > den) m = den - m;            // sin(pi - x) = sin x: m in [0, den/2]
  return sign * sin(M_PI * ((double)m / (double)den));
}

// Morse spectrum sample A(theta) for the scale with peak omega:
//   2 exp(-beta ln w0 + w0^gamma + beta ln w - w^gamma),  w = theta w0 / omega
// (morseutils.py:116-117, :130-131; 'bandpass' normalisation, first family: peak value 2)
__host__ __device__ inline double morse_amplitude(double theta, double omega, double gamma,
                                                  double beta, double w0) {
  if (!(theta > 0.0)) return 0.0;
  const double w = theta * (w0 / omega);
  return 2.0 * exp(-beta * log(w0) + pow(w0, gamma) + beta * log(w) - pow(w, gamma));
}

// G(2 pi a / b) for a kernel of L taps whose kept spectrum samples are amp[0..n_bins) at
// bins bin_lo .. bin_lo + n_bins - 1.
__host__ __device__ inline double exact_gain(const double* amp, int32_t bin_lo, int32_t n_bins,
                                             int64_t L, int64_t a, int64_t b) {
  const int64_t lb = L * b;
  double acc = 0.0;
  for (int32_t i = 0; i < n_bins; ++i) {
    // theta_j - theta = 2 pi (j b - a L) / (L b)
    const int64_t num = (int64_t)(bin_lo + i) * b - a * L;
    double dir;
    if (num % lb == 0) {                       // theta = theta_j (mod 2 pi): the limit
      const int64_t m = num / lb;
      dir = ((m * (L - 1)) & 1) ? -(double)L : (double)L;
    } else {
      dir = sinpi_ratio(num, b) / sinpi_ratio(num, lb);
    }
    acc += amp[i] * dir;
  }
  return acc / (double)L;
}

}  // namespace gcwt
