// planner.cpp -- see planner.h
#include "planner.h"
#include "interp.h"
#include "morse_exact.h"
#include "options.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cmath>
#include <map>
#include <atomic>
#include <mutex>
#include <new>
#include <thread>

namespace gcwt {

// Decimated samples a block gives up at each edge beyond the kernel's measured support.  One is
// structural (a phase r/R of the synthesis looks up to one decimated sample past its block
// position); the second is slack.  Option halo_margin overrides (measure build only).
// The three ways to convolve with a kernel no decimated band holds, by its length (planner.h)
static int exact_method(const HostPlan& hp, int64_t length) {
  if (length <= hp.direct_max_len) return GCWT_SCALE_DIRECT;
  if (length <= hp.blockconv_max_len) return GCWT_SCALE_BLOCKCONV;
  return GCWT_SCALE_FULLBAND;
}
static bool uses_segment_fft(int method) { return method == GCWT_SCALE_SPECTRAL || method == GCWT_SCALE_FULLBAND; }

// Block convolution: the scales sorted by kernel length and cut into groups.  A group shares the spectra of its
// blocks, so its hop is set by its longest kernel -- sample j of a 4096-sample block is good for the scale with
// `b` taps behind the output sample and `f` ahead of it when b <= j < 4096 - f -- and a block costs one forward
// transform (float64, two blocks each, its own launch: about two of the others) plus one inverse per scale: the
// cut that minimises sum (2 + scales) / hop, by dynamic programming over the sorted list.  With `rebalance` the
// shortest kernels may go back to the time domain where that is cheaper: 39 taps there cost what a scale of a
// full block of the widest hop does (6.4 us per tap against 0.25 ms per scale: 128 ch x 1e6 on an MI355X), a
// group of two or three scales does not pay for its spectra, and blocks that epochs fill to `bc_fill` cost the same
// as full ones.
static void plan_blockconv(HostPlan* hp, bool rebalance) {
  std::vector<int>& order = hp->bc_order;
  order.clear();
  for (size_t i = 0; i < hp->scales.size(); ++i)
    if (hp->scales[i].method == GCWT_SCALE_BLOCKCONV) order.push_back((int)i);
  hp->bc_groups.clear();
  std::stable_sort(order.begin(), order.end(),
                   [&](int x, int y) { return hp->scales[x].length < hp->scales[y].length; });
  const int n = (int)order.size();
  // the longest kernel of a run decides both sides: ahead (L - 1) / 2 taps, behind L - 1 - (L - 1) / 2, the latter
  // rounded up to 64 so that a block's stores start on a 256-byte boundary of the row
  const int ramp = hp->exact_only ? kBlockConvRamp : 0;
  auto geometry = [&](int last, int* hop, int* back) {
    const int64_t len = hp->scales[order[last]].length;
    const int ahead = (int)((len - 1) / 2);
    *back = (int)(((len - 1 - ahead) + ramp + 63) & ~(int64_t)63);
    *hop = (kRowLen - *back - ahead - ramp) & ~63;
  };
  const double kForward = 2.0, kWidestHop = 3712.0, kTapsPerUnit = 39.0;
  std::vector<double> rest(n + 1, 0.0);        // cheapest grouping of order[start ..), in scales of a full widest block
  std::vector<int> stop(n + 1, n);
  for (int start = n - 1; start >= 0; --start) {
    rest[start] = 1e300;
    for (int end = start + 1; end <= n; ++end) {
      int hop, back;
      geometry(end - 1, &hop, &back);
      const double c = (kForward + (end - start)) * kWidestHop / (double)hop + rest[end];
      if (c < rest[start]) { rest[start] = c; stop[start] = end; }
    }
  }
  int first = 0;
  if (rebalance) {
    double taps = 0.0, best = rest[0] / hp->bc_fill;
    for (int k = 1; k <= n && hp->scales[order[k - 1]].length <= kDirectMaxLen; ++k) {
      taps += (double)hp->scales[order[k - 1]].length;
      const double c = taps / kTapsPerUnit + rest[k] / hp->bc_fill;
      if (c < best) { best = c; first = k; }
    }
    for (int k = 0; k < first; ++k) hp->scales[order[k]].method = GCWT_SCALE_DIRECT;
  }
  for (int start = first; start < n; start = stop[start]) {
    HostPlan::BcGroup g;
    g.first = start - first;
    g.count = stop[start] - start;
    geometry(stop[start] - 1, &g.hop, &g.back);
    g.ramp = ramp;
    hp->bc_groups.push_back(g);
  }
  order.erase(order.begin(), order.begin() + first);
  hp->n_blockconv = (int)order.size();
  for (int k = 0; k < hp->n_blockconv; ++k) hp->scales[order[k]].blockconv_index = k;
}

static int halo_margin() { return (int)std::max<long long>(0, option_or("halo_margin", 2)); }

double morse_log_gain(double u, double gamma, double beta) {
  return beta * std::log(u) - (beta / gamma) * (std::pow(u, gamma) - 1.0);
}

static void band_edges(double gamma, double beta, double eps, double* u_lo, double* u_hi) {
  const double target = std::log(eps);
  double lo = 1e-9, hi = 1.0;
  for (int i = 0; i < 200; ++i) {
    double mid = 0.5 * (lo + hi);
    if (morse_log_gain(mid, gamma, beta) < target) lo = mid; else hi = mid;
  }
  *u_lo = lo;
  lo = 1.0; hi = 64.0;
  for (int i = 0; i < 200; ++i) {
    double mid = 0.5 * (lo + hi);
    if (morse_log_gain(mid, gamma, beta) < target) hi = mid; else lo = mid;
  }
  *u_hi = hi;
}

// Bins of the reference's L-point spectrum grid that matter: A_j above 1e-18 of the peak,
// among the kept bins 1 .. round(L/2)-1 (morseutils.py:178; Python's round: half to even).
static int64_t round_half_even_half(int64_t L) {   // round(L / 2)
  if (L % 2 == 0) return L / 2;
  const int64_t lo = L / 2;
  return (lo % 2 == 0) ? lo : lo + 1;
}

// Spectrum sample j of the L-point grid for any member of the family (morseutils.py:115-133,
// :175-196): psizero(w) * coeff * Laguerre_k^{(c)}(2 w^gamma), w = theta_j w0 / omega.
//   'bandpass': psizero = 2 exp(-b ln w0 + w0^g + b ln w - w^g), coeff = sqrt(exp(lgamma(r) +
//               lgamma(k+1) - lgamma(k+r))) (1 for beta = 0)
//   'energy':   psizero = exp(b ** ln(w) - w^g) -- the reference's expression as written
//               (:124), its numbers are the contract -- coeff = sqrt(omega... 1/fact) * A_k
// psizero(0) is halved (:133).
static double family_sample(int64_t j, int64_t L, double omega, double g, double b, double w0,
                            int order, bool energy) {
  const double fact = omega / w0;
  const double w = 2.0 * M_PI * ((double)j / (double)L) / fact;
  double psizero;
  if (energy) {
    psizero = b == 0.0 ? std::exp(-std::pow(w, g)) : std::exp(std::pow(b, std::log(w)) - std::pow(w, g));
  } else {
    psizero = b == 0.0 ? 2.0 * std::exp(-std::pow(w, g))
                       : 2.0 * std::exp(-b * std::log(w0) + std::pow(w0, g) + b * std::log(w) - std::pow(w, g));
  }
  if (j == 0) psizero *= 0.5;
  if (!std::isfinite(psizero)) psizero = 0.0;                 // morseutils.py:142
  const double r = (2.0 * b + 1.0) / g, c = r - 1.0;
  double coeff = 1.0;
  if (energy) {
    const double a = std::sqrt(2.0 * M_PI * g * std::pow(2.0, r) *
                               std::exp(std::lgamma((double)order + 1.0) - std::lgamma((double)order + r)));
    coeff = std::sqrt(1.0 / fact) * a;                        // morseutils.py:186-189, :249-251
  } else if (b != 0.0) {
    coeff = std::sqrt(std::exp(std::lgamma(r) + std::lgamma((double)order + 1.0) - std::lgamma((double)order + r)));
  }
  const double x = 2.0 * std::pow(w, g);
  double lag = 0.0, xm = 1.0;                                 // generalized Laguerre (morseutils.py:256-273)
  for (int m = 0; m <= order; ++m) {
    const double f = std::exp(std::lgamma(order + c + 1.0) - std::lgamma(c + m + 1.0) - std::lgamma(order - m + 1.0));
    lag += ((m & 1) ? -1.0 : 1.0) * f * xm / std::tgamma(m + 1.0);
    xm *= x;
  }
  const double v = coeff * psizero * lag;
  return std::isfinite(v) ? v : 0.0;
}

static void scale_bins(const HostPlan& hp, ScalePlan* sp, std::vector<double>* amp) {
  const int64_t L = sp->length, K = round_half_even_half(L);
  amp->clear();
  const int flags = hp.prm.wavelet_flags;
  if (flags != 0) {
    // other family members: every kept bin, then the negligible ends trimmed.  A 'bandpass'
    // member is the first wavelet times a degree-k polynomial in w^gamma, so it is negligible
    // wherever the first wavelet is below 1e-40 of its peak; 'energy' members (a different
    // envelope, non-zero at zero frequency) are evaluated on every kept bin.
    const int order = flags & 0xff;
    const bool energy = (flags & GCWT_WAVELET_ENERGY) != 0;
    int64_t j_first = 0, j_last = K - 1;
    if (!energy) {
      double ulo, uhi;
      band_edges(hp.prm.gamma, hp.prm.beta, 1e-40, &ulo, &uhi);
      const double per_u = sp->omega * (double)L / (2.0 * M_PI);
      j_first = std::max<int64_t>(0, (int64_t)std::floor(ulo * per_u) - 1);
      j_last = std::min<int64_t>(K - 1, (int64_t)std::ceil(uhi * per_u) + 1);
    }
    std::vector<double> all((size_t)std::max<int64_t>(K, 0), 0.0);
    double top = 0.0;
    for (int64_t j = j_first; j <= j_last; ++j) {
      all[(size_t)j] = family_sample(j, L, sp->omega, hp.prm.gamma, hp.prm.beta, hp.w0, order, energy);
      top = std::max(top, std::fabs(all[(size_t)j]));
    }
    int64_t lo = 0, hi = K - 1;
    while (lo <= hi && std::fabs(all[(size_t)lo]) <= 1e-18 * top) ++lo;
    while (hi >= lo && std::fabs(all[(size_t)hi]) <= 1e-18 * top) --hi;
    sp->bin_lo = (int32_t)lo;
    if (hi >= lo) amp->assign(all.begin() + lo, all.begin() + hi + 1);
    sp->n_bins = (int32_t)amp->size();
    return;
  }
  const double per_u = sp->omega * (double)L / (2.0 * M_PI);   // bins per unit of u = theta/omega
  int64_t lo = std::max<int64_t>(1, (int64_t)std::floor(hp.u_lo * per_u));
  int64_t hi = std::min<int64_t>(K - 1, (int64_t)std::ceil(hp.u_hi * per_u));
  sp->bin_lo = (int32_t)lo;
  for (int64_t j = lo; j <= hi; ++j)
    amp->push_back(morse_amplitude(2.0 * M_PI * (double)j / (double)L, sp->omega, hp.prm.gamma,
                                   hp.prm.beta, hp.w0));
  sp->n_bins = (int32_t)amp->size();
}

// Measures, on the exact response G of the reference's kernel (morse_exact.h):
//  * theta_hi, theta_neg: outside [-theta_neg, theta_hi] |G| stays below band_tol of the peak.
//    G is probed where its side lobes peak, half-way between grid bins: every half-bin of the
//    upper skirt, then geometric steps out to Nyquist and, from the other end, down from zero
//    frequency -- the side lobes of the L-tap truncation decay like 1/|theta - band|, smoothly,
//    on both sides.  The default wavelet has none below zero (theta_neg = 0); heavy-tailed ones
//    do, largest just below zero frequency, and take a level band shifted to hold them.
//  * support: the distance from the kernel's centre beyond which less than support_tol of
//    its energy (L2) lies.  |psi| is the envelope of an analytic signal, so it is smooth
//    and sampled on a coarse grid from the kernel's end inwards.
static void analyse_scale(const HostPlan& hp, ScalePlan* sp, const double* amp) {
  const int64_t L = sp->length;
  const int32_t nb = sp->n_bins, j0 = sp->bin_lo;
  sp->band_ok = false;
  sp->theta_hi = 2.0 * M_PI;
  sp->theta_neg = 0.0;
  sp->support = 0.5 * (double)L;
  if (nb == 0) return;
  double pk = 0.0, energy = 0.0;
  int32_t i_pk = 0;
  for (int32_t i = 0; i < nb; ++i) {
    if (std::fabs(amp[i]) > pk) { pk = std::fabs(amp[i]); i_pk = i; }
    energy += amp[i] * amp[i];
  }
  const double lim = hp.band_tol * pk;
  auto env = [&](int64_t m) {   // |G| at half-bin m + 1/2: theta = 2 pi (2m + 1) / (2 L)
    return std::fabs(exact_gain(amp, j0, nb, L, 2 * m + 1, 2 * L));
  };
  // highest probed half-bin with |G| above the limit, on the positive side; on the negative side
  // (theta = -2 pi (o - 1/2) / L, i.e. half-bin L - o) the largest such offset o.  Beyond the
  // skirt the probes step geometrically: the extent is taken to the first probe BELOW the limit
  // after the last one above it.
  int64_t top = -1, neg = 0;
  const int64_t m_pk = j0 + i_pk, m_skirt = std::min<int64_t>(L - 1, (int64_t)j0 + nb + 2);
  // (every half-bin of the skirt -- unless the kernel keeps thousands of bins ('energy' members with
  // very long kernels: each probe costs n_bins evaluations), then about a thousand of them)
  const int64_t skirt_step = std::max<int64_t>(1, (int64_t)nb / 1024);
  for (int64_t m = m_pk; m <= m_skirt; m += skirt_step)
    if (env(m) > lim) top = std::min<int64_t>(m + skirt_step - 1, m_skirt);
  bool pos_to_nyquist = false, neg_to_nyquist = false;
  for (int side = 0; side < 2 && m_skirt < L - 1; ++side) {
    double step = 1.0;
    bool pending = false;             // the previous probe was above the limit
    for (double off = 1.0; ; off += step, step *= 1.12) {
      const int64_t o = (int64_t)off;
      const int64_t m = side == 0 ? m_skirt + o : L - o;
      if (side == 0 && m > L / 2) { if (pending) pos_to_nyquist = true; break; }
      if (side == 1 && (m <= L / 2 || m <= m_skirt)) { if (pending) neg_to_nyquist = true; break; }
      const bool above = env(m) > lim;
      if (side == 0) {
        if (above) top = std::max(top, m);
        else if (pending) top = std::max(top, m);   // the edge lies between the two probes
      } else {
        if (above || pending) neg = std::max(neg, o);
      }
      pending = above;
    }
  }
  // (a response that still matters at Nyquist wraps round: top >= L / 2 gives theta_hi > pi, no decimation)
  if (pos_to_nyquist) top = std::max(top, L / 2);
  sp->theta_hi = 2.0 * M_PI * (double)(top + 1) / (double)L;
  sp->theta_neg = neg_to_nyquist ? M_PI : std::min(M_PI, 2.0 * M_PI * (double)neg / (double)L);
  sp->band_ok = sp->theta_hi + sp->theta_neg <= M_PI;
  // theta_lo: from zero frequency up, the half-bins (side-lobe maxima) whose |G| stays below low_tol
  {
    const double low = hp.low_tol * pk;
    int64_t m = 0;
    const int64_t m_stop = std::min<int64_t>(m_pk, 4096);
    while (m < m_stop && env(m) <= low) ++m;
    sp->theta_lo = sp->theta_neg > 0.0 ? 0.0 : 2.0 * M_PI * (double)m / (double)L;
  }
  // A kernel that answers above low_tol all the way down to zero frequency gets no low cut, so whatever a recording
  // carries down there reaches the block convolution -- and the tails a block halo would cut off (support_tol is an
  // energy figure: right for a white input) answer to it in full: 8.5e-6 on Morse(6, 13.2) with a drift of 10 x the
  // recording's spread, in float64 (round 4's soak).  Such kernels keep their whole length (three more decimated
  // samples of halo for that one; the default wavelet, whose levels are cut, is not concerned).
  if (hp.high_precision && !(sp->theta_lo > 0.0)) { sp->support = 0.5 * (double)L; return; }

  // |psi(centre + t)|^2 = |(1/L) sum_j A_j e^{i theta_j t}|^2, same on both sides (A real);
  // total energy (1/L) sum A_j^2 (Parseval).  Walk in from t = L/2 until the tails hold
  // support_tol^2 of it.
  const int kProbes = 48;
  const double h = 0.25 * (double)L / kProbes, total = energy / (double)L;
  double tail = 0.0;
  sp->support = 0.25 * (double)L;
  for (int q = 0; q < kProbes; ++q) {
    const double t = 0.5 * (double)L - ((double)q + 0.5) * h;
    double re = 0.0, im = 0.0;
    for (int32_t i = 0; i < nb; ++i) {
      const double ph = 2.0 * M_PI * std::fmod((double)(j0 + i) * t / (double)L, 1.0);
      re += amp[i] * std::cos(ph);
      im += amp[i] * std::sin(ph);
    }
    tail += 2.0 * h * (re * re + im * im) / ((double)L * (double)L);
    if (tail > hp.support_tol * hp.support_tol * total) {
      sp->support = std::min(0.5 * (double)L, t + 0.5 * h);
      break;
    }
  }
}

// Decides whether a level is made by the interpolating synthesis and, if so, designs it: the
// demodulation bin of every scale (the centre of the bins where its gain exceeds 1e-4 of the
// peak, a multiple of q), the envelope of the level's gains against the distance from that
// centre, the coefficient tables of both kernel-length parities fitted under that envelope
// (interp.h: design_interp_weighted), and -- from the scales' own gains -- a bound on the error
// the interpolation adds: max over bins of |G_s[k]| / peak times the interpolator's error at that
// bin's distance from the demodulation centre.  A level whose bound exceeds interp_tol stays on
// the FFT-per-sample kernels.
// One candidate: q phases per (block, scale) through the block transform, T taps between them (6 or 8; the table
// rows are kInterpTaps wide either way, the six in the middle).  true: designed and within the tolerance.
static bool design_interp_level(HostPlan* hp, LevelPlan* lp, int q, int T, std::vector<float>* coef_out) {
  const int B = hp->block, R = lp->decimation;
  if (R < 4 * q) return false;                             // I = R / q >= 4: a lane makes 4 samples of one interval
  while (R / q > kInterpMaxFactor) q *= 2;
  if (q > kInterpMaxPhases) return false;                  // a pass of the kernel has 16 columns
  const int I = R / q;
  for (int sidx : lp->scales)
    if (hp->scales[sidx].n_bins > 4096) return false;            // 'energy' members with very long kernels: the gain
                                                           // probes below would take seconds
  // gains on the level's grid, band edges and demodulation bins
  std::vector<std::vector<double>> gains(lp->scales.size(), std::vector<double>((size_t)B));
  std::vector<std::vector<double>> gains_all(lp->scales.size(), std::vector<double>((size_t)B));   // for the bound: every bin
  std::vector<double> genv((size_t)2 * B, 0.0);            // envelope against the distance d from the centre: index d + B
  for (size_t n = 0; n < lp->scales.size(); ++n) {
    ScalePlan& sp = hp->scales[lp->scales[n]];
    std::vector<double>& g = gains[n];
    double pk = 0.0;
    // (where the kept spectrum samples are more than a few bins of the L-point grid away the response
    // is side lobes of the truncation, below 1e-6 of the peak for every wavelet that is decimated at
    // all: zero for the purposes of this design, and two thirds of the evaluations saved)
    const double bin_per_k = (double)sp.length / ((double)B * R);        // L-grid bins per level bin
    const double lo_k = ((double)sp.bin_lo - 6.0) / bin_per_k + lp->band_shift;
    const double hi_k = ((double)(sp.bin_lo + sp.n_bins) + 6.0) / bin_per_k + lp->band_shift;
    for (int k = 0; k < B; ++k) {
      const double ge = std::fabs(exact_gain(hp->amps.data() + sp.amp_offset, sp.bin_lo, sp.n_bins, sp.length,
                                             k - lp->band_shift, (int64_t)B * R));
      gains_all[n][(size_t)k] = ge;
      g[(size_t)k] = ((double)k < lo_k || (double)k > hi_k) ? 0.0 : ge;    // the design sees the band and its skirt
      pk = std::max(pk, g[(size_t)k]);
    }
    if (!(pk > 0.0)) return false;
    int klo = B, khi = -1;
    for (int k = 0; k < B; ++k) {
      gains_all[n][(size_t)k] /= pk;
      g[(size_t)k] /= pk;
      if (g[(size_t)k] > 1e-4) { klo = std::min(klo, k); khi = std::max(khi, k); }
    }
    int kc = ((klo + khi + q) / (2 * q)) * q;              // nearest multiple of q to the middle
    kc = std::min(std::max(kc, 0), B - q);
    sp.demod_bin = kc;
    for (int k = 0; k < B; ++k) genv[(size_t)(k - kc + B)] = std::max(genv[(size_t)(k - kc + B)], g[(size_t)k]);
  }
  for (int d = 1; d < B; ++d) {                            // real coefficients: the error is even in d
    const double m = std::max(genv[(size_t)(B + d)], genv[(size_t)(B - d)]);
    genv[(size_t)(B + d)] = genv[(size_t)(B - d)] = m;
  }
  double hw = 0.0;
  for (int d = 0; d < B; ++d)
    if (genv[(size_t)(B + d)] > 1e-4) hw = (double)d + 1.0;
  const double alpha = hw / (0.5 * B) / (double)q;
  if (alpha > 0.9) return false;
  // The same level geometry comes back with every plan of a recording (one transform() per call):
  // a small process-wide cache keyed by what the design depends on -- the envelope, to float
  // precision -- saves the fit (the bound below is recomputed: it depends on the scales).
  std::vector<double> c((size_t)2 * I * T);
  {
    static std::mutex mu;
    static std::map<std::vector<float>, std::vector<double>> cache;
    std::vector<float> key{(float)T, (float)I, (float)q, (float)B};
    for (double v : genv) key.push_back((float)v);
    bool hit = false;
    {
      std::lock_guard<std::mutex> lock(mu);
      auto it = cache.find(key);
      if (it != cache.end()) { c = it->second; hit = true; }
    }
    if (!hit) {                           // (outside the lock: the levels of a plan are designed side by side)
      design_interp_weighted(T, I, q, B, genv.data(), 0.0, c.data());
      design_interp_weighted(T, I, q, B, genv.data(), 0.5, c.data() + (size_t)I * T);
      std::lock_guard<std::mutex> lock(mu);
      if (cache.size() >= 64) cache.clear();
      cache.emplace(std::move(key), c);
    }
  }
  // the tables the kernel uses are float32: bound the error with what it will multiply by
  for (double& v : c) v = (double)(float)v;
  // interpolator error against the distance d from the demodulation centre, worst over the sub-sample
  // positions of both parities: every one of them up to I = 64, 64 evenly spaced ones beyond (with a margin:
  // round 3 probed 8 and read up to 2.3 x too low at I = 1024), at EVERY distance -- the side lobes outside the
  // design band count with their own (tiny) gains
  std::vector<double> err((size_t)2 * B, 0.0);             // d = -B .. B-1 at index d + B
  const int n_probe = std::min(I, 64);
  const double probe_margin = n_probe < I ? 1.25 : 1.0;
  for (int par = 0; par < 2; ++par)
    for (int pr = 0; pr < n_probe; ++pr) {
      const int rho = (int)(((int64_t)(2 * pr + 1) * I) / (2 * n_probe));
      const double tau = ((double)rho - 0.5 * par) / (double)I;
      const double* cr = c.data() + ((size_t)par * I + rho) * T;
      for (int d = -B; d < B; ++d) {
        const double th = std::remainder(2.0 * M_PI * (double)d / ((double)B * q), 2.0 * M_PI);
        err[(size_t)(d + B)] = std::max(err[(size_t)(d + B)], interp_error_at(T, cr, tau, th));
      }
    }
  double bound = 0.0;
  for (size_t n = 0; n < lp->scales.size(); ++n) {
    const int kc = hp->scales[lp->scales[n]].demod_bin;
    for (int k = 0; k < B; ++k) bound = std::max(bound, gains_all[n][(size_t)k] * err[(size_t)(k - kc + B)]);
  }
  bound *= probe_margin;
  lp->interp_err = bound;
  lp->interp_alpha = alpha;
  if (!(bound <= hp->interp_tol)) return false;
  lp->interp_q = q;
  lp->interp_taps = T;
  lp->interp_factor = I;
  lp->coef_offset = 0;                    // within coef_out: the caller places it in HostPlan::interp_coef
  // rows of kInterpTaps floats: tap j of T sits at j + (kInterpTaps - T) / 2, i.e. at the same z sample
  coef_out->clear();
  const int pad = (kInterpTaps - T) / 2;
  for (size_t row = 0; row < (size_t)2 * I; ++row)
    for (int j = 0; j < kInterpTaps; ++j)
      coef_out->push_back(j >= pad && j < pad + T ? (float)c[row * T + (size_t)(j - pad)] : 0.f);
  return true;
}

static void plan_interp_level_uncached(HostPlan* hp, LevelPlan* lp, std::vector<float>* coef_out) {
  const int R = lp->decimation;
  lp->interp_q = 0;
  if (hp->prm.out_mode == GCWT_OUT_COMPLEX_C64) return;   // the demodulation would have to be undone per sample
  // Below R = 16 the q = 2 phases through the block transform are a quarter or more of the
  // FFT-per-sample work and the interpolation does not pay (profiles/r03_synth_study.md);
  // GHOSTCWT_INTERP_MIN_R moves the line, GHOSTCWT_INTERP_Q / _TAPS fix the phases per scale and the taps (A/B runs).
  const int min_r = (int)std::max<long long>(8, option_or("interp_min_r", 16));     // (measure build only)
  if (R < min_r || lp->scales.empty()) return;
  if (lp->scales.size() > 256) return;                     // the kernel parks a level's scale list in LDS
  // From R = 32 up four phases and six taps: a quarter fewer multiply-adds per stored sample for twice the (small)
  // transform work, and the closer spacing makes the shorter filter the more accurate one (profiles/r05_synth_study.md
  // 5); eight taps, then two phases, where the bound says otherwise.  The pipelined kernel (option synthp) is written for
  // two phases and eight taps.
  const long long oq = option_or("interp_q", 0), ot = option_or("interp_taps", 0);
  std::vector<std::pair<int, int>> cands;
  if (oq || ot) cands.push_back({oq == 4 ? 4 : 2, ot == 6 ? 6 : 8});
  else if (R < 32 || option_or("synthp", 0) != 0) cands.push_back({2, 8});
  else cands = {{4, 6}, {4, 8}, {2, 8}};
  for (const auto& cd : cands)
    if (design_interp_level(hp, lp, cd.first, cd.second, coef_out)) return;
}

// The same levels come back with every plan of a recording (ContinuousWaveletTransform.transform() makes one per
// call) and a level's design -- the gains on 256 bins of every scale, the fit, the bound over every sub-sample
// position -- is most of a plan's host time (20 of 29 ms for the headline grid): the whole outcome is kept,
// process-wide, keyed by everything it depends on.
namespace {
struct InterpDesign {
  int q, taps, factor;
  double alpha, err;
  std::vector<int> demod;
  std::vector<float> coef;       // empty when the level is not interpolated
};
}  // namespace

static void plan_interp_level(HostPlan* hp, LevelPlan* lp, std::vector<float>* coef_out) {
  static std::mutex mu;
  static std::map<std::vector<double>, InterpDesign> cache;
  std::vector<double> key{(double)hp->block, (double)lp->decimation, (double)lp->band_shift, (double)hp->prm.out_mode,
                          hp->prm.gamma, hp->prm.beta, (double)hp->prm.wavelet_flags, hp->interp_tol,
                          (double)option_or("interp_min_r", 16), (double)option_or("interp_q", 0),
                          (double)option_or("interp_taps", 0), (double)option_or("synthp", 0)};
  for (int sidx : lp->scales) { key.push_back(hp->scales[sidx].omega); key.push_back((double)hp->scales[sidx].length); }
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
      const InterpDesign& d = it->second;
      lp->interp_q = d.coef.empty() ? 0 : d.q;
      lp->interp_taps = d.taps;
      lp->interp_factor = d.factor;
      lp->interp_alpha = d.alpha;
      lp->interp_err = d.err;
      for (size_t n = 0; n < lp->scales.size() && n < d.demod.size(); ++n) hp->scales[lp->scales[n]].demod_bin = d.demod[n];
      if (!d.coef.empty()) {
        lp->coef_offset = 0;
        *coef_out = d.coef;
      }
      return;
    }
  }
  plan_interp_level_uncached(hp, lp, coef_out);
  InterpDesign d{lp->interp_q, lp->interp_taps, lp->interp_factor, lp->interp_alpha, lp->interp_err, {}, {}};
  for (int sidx : lp->scales) d.demod.push_back(hp->scales[sidx].demod_bin);
  if (lp->interp_q > 0) d.coef = *coef_out;
  else coef_out->clear();
  std::lock_guard<std::mutex> lock(mu);
  if (cache.size() >= 256) cache.clear();
  cache.emplace(std::move(key), std::move(d));
}

// samples over which a time block's own edges fade out beyond the halo (precision = high): a quarter of the halo, i.e.
// an eighth of the longest kernel -- (2 pi / (theta T))^3 of a hard cut's leakage at theta: 1e-10 at the top of config
// 5's grid, 0.2 at its lowest scale, whose band holds most of a steep recording's power anyway; config 5 keeps its five
// blocks of 2^22 samples per 18e6 (half the halo would make it six)
static int64_t block_ramp(int64_t halo_s) { return (halo_s / 4 + 63) & ~(int64_t)63; }

static int64_t next_pow2(int64_t v) {
  int64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

int build_host_plan(const gcwt_params& prm, HostPlan* hp, std::string* err) {
  auto fail = [&](int code, const std::string& msg) { *err = msg; return code; };

  if (prm.n_samples <= 0) return fail(GCWT_ERR_INVALID, "n_samples must be positive");
  if (prm.n_channels <= 0) return fail(GCWT_ERR_INVALID, "n_channels must be positive");
  if (prm.n_channels > 65535) return fail(GCWT_ERR_UNSUPPORTED, "n_channels > 65535 per plan");
  if (prm.n_freqs <= 0 || !prm.freqs_hz) return fail(GCWT_ERR_INVALID, "no analysis frequencies");
  if (!(prm.fs > 0) || !std::isfinite(prm.fs)) return fail(GCWT_ERR_INVALID, "Sampling rate must be positive and finite");
  if (!(prm.gamma > 0) || !std::isfinite(prm.gamma)) return fail(GCWT_ERR_INVALID, "gamma must be positive and finite");
  if (!(prm.beta > 0) || !std::isfinite(prm.beta)) return fail(GCWT_ERR_INVALID, "beta must be positive and finite");
  if (prm.wavelet_flags < 0 || (prm.wavelet_flags & ~(0xff | GCWT_WAVELET_ENERGY)) || (prm.wavelet_flags & 0xff) > 32)
    return fail(GCWT_ERR_INVALID, "bad wavelet_flags (order 0..32, GCWT_WAVELET_ENERGY)");
  if (prm.out_mode < 0 || prm.out_mode > 2) return fail(GCWT_ERR_INVALID, "bad out_mode");
  if (prm.block != 0 && prm.block != 256)
    return fail(GCWT_ERR_UNSUPPORTED, "only block = 256 is built");

  hp->prm = prm;
  hp->block = 256;
  hp->band_tol = prm.band_eps > 0 ? prm.band_eps : 2e-7;
  if (hp->band_tol > 1e-3) return fail(GCWT_ERR_INVALID, "band_eps too large");
  if (prm.support_tol > 0) hp->support_tol = prm.support_tol;
  if (hp->support_tol > 1e-2) return fail(GCWT_ERR_INVALID, "support_tol too large");
  if (prm.reserved0 != 0) return fail(GCWT_ERR_INVALID, "gcwt_params.reserved0 must be 0 (caller built against an older ghostcwt.h?)");
  if (prm.precision < 0 || prm.precision > 4) return fail(GCWT_ERR_INVALID, "bad precision (0 default = 4 auto, 1 fast, 2 high, 3 exact)");
  hp->high_precision = prm.precision != GCWT_PRECISION_FAST;
  hp->exact_only = prm.precision == GCWT_PRECISION_EXACT;
  hp->auto_precision = prm.precision == GCWT_PRECISION_DEFAULT || prm.precision == GCWT_PRECISION_AUTO;
  if (hp->exact_only) {                     // every kernel through a float64 spectrum: none in the time domain either
    hp->direct_max_len = 0;
    hp->blockconv_max_len = kBlockConvExactMaxLen;
  } else if (option_or("blockconv", 1) != 0) {
    hp->direct_max_len = (int)std::min<long long>(kDirectMaxLen, std::max<long long>(0, option_or("direct_max_len", kDirectDefaultLen)));
    hp->blockconv_max_len = kBlockConvMaxLen;
  } else {                                  // rounds 1-4: time domain up to 256 taps, full band beyond
    hp->direct_max_len = kDirectMaxLen;
    hp->blockconv_max_len = 0;
  }
  hp->freqs.assign(prm.freqs_hz, prm.freqs_hz + prm.n_freqs);
  hp->out_elem_bytes = prm.out_mode == GCWT_OUT_COMPLEX_C64 ? 8 : 4;

  // epochs
  if (prm.n_epochs <= 0 || !prm.epoch_bounds) {
    hp->bounds = {0, prm.n_samples};
  } else {
    hp->bounds.assign(prm.epoch_bounds, prm.epoch_bounds + 2 * (size_t)prm.n_epochs);
  }
  const int n_ep = (int)hp->bounds.size() / 2;
  for (int e = 0; e < n_ep; ++e) {
    int64_t s = hp->bounds[2 * e], t = hp->bounds[2 * e + 1];
    if (s < 0 || t > prm.n_samples || t <= s)
      return fail(GCWT_ERR_INVALID, "epoch bounds outside the data or empty");
  }
  if (n_ep > 1) {
    // epochs are disjoint runs of samples (utils.py:3-42 cuts the recording where its clock jumps): two that share a
    // sample would both write it
    std::vector<std::pair<int64_t, int64_t>> runs;
    for (int e = 0; e < n_ep; ++e) runs.push_back({hp->bounds[2 * e], hp->bounds[2 * e + 1]});
    std::sort(runs.begin(), runs.end());
    for (int e = 1; e < n_ep; ++e)
      if (runs[e].first < runs[e - 1].second) return fail(GCWT_ERR_INVALID, "epoch bounds overlap");
  }
  {
    // how full the 4096-sample blocks of the block convolution would be: a block costs the same however little
    // of an epoch it holds (plan_blockconv weighs that against the time domain)
    int64_t blocks = 0, samples = 0;
    for (int e = 0; e < n_ep; ++e) {
      const int64_t len = hp->bounds[2 * e + 1] - hp->bounds[2 * e];
      samples += len;
      blocks += (len + 3711) / 3712;
    }
    hp->bc_fill = (double)samples / (3712.0 * (double)std::max<int64_t>(1, blocks));
  }

  const double g = prm.gamma, b = prm.beta;
  hp->w0 = std::exp((std::log(b) - std::log(g)) / g);                 // morseutils.py:315
  hp->base_length = (2.0 * std::sqrt(2.0) * std::sqrt(g * b)) / hp->w0 * 4.0;  // morse.py:115
  band_edges(g, b, 1e-18, &hp->u_lo, &hp->u_hi);

  // Per scale: the reference's kernel length, its kept spectrum samples, and what the
  // exact response of that kernel allows (analyse_scale).  A scale is
  //   SPECTRAL  when its response is below band_tol on [theta_hi, 2 pi) with
  //             theta_hi <= pi -- the band fits a decimation R >= 2 -- and its support
  //             fits the 256-sample decimated block;
  //   DIRECT    otherwise, when the kernel is short (the top of the default grid, where
  //             the filter reaches Nyquist: L <= 52 for gamma, beta = 3, 20);
  //   FULLBAND  otherwise: one FFT convolution over the whole band -- wavelets whose
  //             L-tap truncation leaks everywhere (small beta), any length.
  const int B = hp->block;
  hp->scales.resize(prm.n_freqs);
  std::vector<double> amp;
  for (int i = 0; i < prm.n_freqs; ++i) {
    ScalePlan& sp = hp->scales[i];
    sp.freq_hz = hp->freqs[i];
    // (the reference clamps its grid to the wavelet's range, transforms.py:412-431 -- at most 0.39 fs for the default
    // wavelet; beyond Nyquist a sampled kernel means nothing, and inf / nan would plan a one-tap "kernel")
    if (!(sp.freq_hz > 0) || !std::isfinite(sp.freq_hz)) return fail(GCWT_ERR_INVALID, "analysis frequencies must be positive and finite");
    if (sp.freq_hz > 0.5 * prm.fs) return fail(GCWT_ERR_INVALID, "analysis frequency above the Nyquist frequency fs / 2");
    sp.omega = sp.freq_hz / (prm.fs / 2.0) * M_PI;                    // transforms.py:408-410
    sp.length = (int64_t)std::ceil(hp->w0 / sp.omega * hp->base_length);  // morse.py:118-122
    if (sp.length < 1) sp.length = 1;
    if (sp.length > ((int64_t)1 << 40)) return fail(GCWT_ERR_UNSUPPORTED, "kernel length beyond 2^40");
    sp.half_delay = (double)(sp.length - 1) / 2.0 - (double)((sp.length - 1) / 2);
    scale_bins(*hp, &sp, &amp);
    sp.amp_offset = (int64_t)hp->amps.size();
    hp->amps.insert(hp->amps.end(), amp.begin(), amp.end());
    hp->max_bins = std::max(hp->max_bins, (int)sp.n_bins);
    analyse_scale(*hp, &sp, amp.data());
    if (hp->exact_only) sp.band_ok = false;
    sp.method = sp.band_ok ? GCWT_SCALE_SPECTRAL
                           : exact_method(*hp, sp.length);
  }

  // Decimation of every spectral candidate needs the shortest FFT of the plan, which needs
  // the longest kernel that goes through an FFT: two rounds (a candidate that turns out not
  // to fit its block moves to the other paths and may shorten the FFTs).
  hp->max_fft_log2 = prm.max_fft_log2 == 0 ? 22 : prm.max_fft_log2;
  if (hp->max_fft_log2 < 12 || hp->max_fft_log2 > kMaxFftLog2)
    return fail(GCWT_ERR_INVALID, "max_fft_log2 must be 0 or 12..24");
  if (prm.max_fft_log2 == 0) {
    // The default stays at 2^22 unless the longest kernel that goes through an FFT does not leave a time block of
    // that length a quarter of it as its core: then 2^23 / 2^24 (long mode), so that every frequency the
    // reference's compute_freq_bounds permits runs (morse.py:93-106: 0.116 Hz for 18e6 samples at 30 kHz)
    int64_t lmax0 = 1;
    for (const ScalePlan& sp : hp->scales)
      if (uses_segment_fft(sp.method)) lmax0 = std::max(lmax0, sp.length);
    const int64_t halo0 = lmax0 / 2 + 2, ramp0 = hp->high_precision ? block_ramp(halo0) : 0;
    while (hp->max_fft_log2 < kMaxFftLog2 &&
           ((((int64_t)1 << hp->max_fft_log2) - 2 * halo0 - 2 * ramp0 - 64) & ~(int64_t)63) < ((int64_t)1 << hp->max_fft_log2) / 4)
      ++hp->max_fft_log2;
  }
  const int64_t pmax = (int64_t)1 << hp->max_fft_log2;
  auto fft_of = [&](int64_t e0, int64_t e1, int64_t lmax) {
    return std::max<int64_t>(kRowLen, next_pow2(e1 - (e0 & ~(int64_t)63) + lmax));
  };
  auto longest_fft_kernel = [&]() {
    int64_t l = 1;
    for (const ScalePlan& sp : hp->scales)
      if (uses_segment_fft(sp.method)) l = std::max(l, sp.length);
    return l;
  };
  for (int round = 0; round < 4; ++round) {
    const int64_t lmax = longest_fft_kernel();
    int64_t pmin = INT64_MAX;
    for (int e = 0; e < n_ep; ++e)
      pmin = std::min(pmin, std::min(pmax, fft_of(hp->bounds[2 * e], hp->bounds[2 * e + 1], lmax)));
    const int r_cap = (int)std::min<int64_t>(kMaxDecimation, pmin / B);
    bool moved = false;
    for (ScalePlan& sp : hp->scales) {
      if (sp.method != GCWT_SCALE_SPECTRAL) continue;
      const bool two_sided = sp.theta_neg > 0.0;
      // does the scale's band fit the 2 pi / r of a level of decimation r?  One-sided: [0, theta_hi].
      // Two-sided: [-theta_neg, theta_hi] inside a band shifted below zero by a whole number of
      // steps of max(1, r / 16) bins (what the spectrum's k1-major layout can slice), so a step
      // and a bin of rounding are set aside; only the two-pass level transforms take a shifted slice.
      auto fits = [&](int r) {
        if (!two_sided) return sp.theta_hi * (double)r <= 2.0 * M_PI;
        if (r > kMaxTwoPassDecimation) return false;
        const int g = std::max(1, r / 16);
        return (sp.theta_hi + sp.theta_neg) * (double)r + 2.0 * M_PI * (double)(g + 1) / (double)B <= 2.0 * M_PI;
      };
      int r = 2;   // one-sided: theta_hi <= pi holds here, so R = 2 always fits
      while (2 * r <= r_cap && fits(2 * r)) r *= 2;
      sp.decimation = r;
      // decimated samples discarded at each block edge: the kernel's measured support
      int halo = std::max((int)std::ceil(sp.support / (double)r) + halo_margin(), 16);
      if (r == 2) halo += halo & 1;   // keeps halo*R a multiple of 4: 16-byte aligned tile runs
      if (B - 2 * halo < 32 || (two_sided && !fits(2))) {   // does not fit the block at the largest decimation it allows
        sp.method = exact_method(*hp, sp.length);
        moved = true;
      }
    }
    // Scales with a two-sided band share their level's shift: the bins below zero are the most
    // any member needs, and every member's upper edge must still fit above them.  Where a
    // decimation's members do not fit together, the one that asks for the most bins steps
    // down to the next lower decimation (twice the room) until they do.
    for (int pass = 0; pass < 64; ++pass) {
      std::map<int, std::vector<int>> by_r;
      for (int i = 0; i < prm.n_freqs; ++i)
        if (hp->scales[i].method == GCWT_SCALE_SPECTRAL) by_r[hp->scales[i].decimation].push_back(i);
      bool changed = false;
      for (auto& kv : by_r) {
        bool any_two_sided = false;
        for (int i : kv.second) any_two_sided = any_two_sided || hp->scales[i].theta_neg > 0.0;
        if (!any_two_sided) continue;          // a plain one-sided level: nothing to agree on
        const int r = kv.first, g = std::max(1, r / 16);
        const double delta = 2.0 * M_PI / ((double)B * r);
        int shift = 0, worst = -1;
        double worst_need = 0.0;
        for (int i : kv.second) shift = std::max(shift, (int)std::ceil(hp->scales[i].theta_neg / delta));
        shift = (shift + g - 1) / g * g;
        for (int i : kv.second) {
          const double need = std::max(hp->scales[i].theta_hi / delta - (double)(B - shift), 0.0);
          if (need > 0.0 && need >= worst_need) { worst_need = need; worst = i; }
        }
        if (worst < 0) continue;
        // the member that takes the most room: the widest band of the group
        int widest = worst;
        for (int i : kv.second)
          if (hp->scales[i].theta_neg + hp->scales[i].theta_hi >
              hp->scales[widest].theta_neg + hp->scales[widest].theta_hi) widest = i;
        ScalePlan& sp = hp->scales[widest];
        changed = true;
        if (r == 2) {
          sp.method = exact_method(*hp, sp.length);
          moved = true;
        } else {
          sp.decimation = r / 2;
          int halo = std::max((int)std::ceil(sp.support / (double)(r / 2)) + halo_margin(), 16);
          if (r / 2 == 2) halo += halo & 1;
          if (B - 2 * halo < 32) {
            sp.method = exact_method(*hp, sp.length);
            moved = true;
          }
        }
        break;                       // regroup after every move
      }
      if (!changed) break;
    }
    if (!moved) break;
  }
  // A decimation whose band is shifted (it holds scales with a two-sided band) goes through k_synth7 only, which
  // parks at most 256 scales of a level: what is beyond that takes the exact full-band (or time-domain) path
  // instead of failing the plan.
  {
    std::map<int, std::vector<int>> by_r;
    for (int i = 0; i < prm.n_freqs; ++i)
      if (hp->scales[i].method == GCWT_SCALE_SPECTRAL) by_r[hp->scales[i].decimation].push_back(i);
    for (auto& kv : by_r) {
      bool shifted = false;
      for (int i : kv.second) shifted = shifted || hp->scales[i].theta_neg > 0.0;
      if (!shifted) continue;
      for (size_t n = 256; n < kv.second.size(); ++n) {
        ScalePlan& sp = hp->scales[kv.second[n]];
        sp.method = exact_method(*hp, sp.length);
      }
    }
  }
  // A decimation that only a few scales reach is folded into the next lower one: every
  // workgroup of a level pays a prologue worth about three scales of its walk (7.8 us against
  // 2.5 us per scale, profiles/r02_synth_study.md 6), so e.g. 3 scales at R = 256 cost as much as
  // 6, while walked by the R = 128 workgroups (block halo 22 -> 25 for all 18 scales) they cost
  // 3.6.  Cost model per level: (3 + scales) / hop.  Only merges that keep the plain block
  // layout (halo <= 32) are taken.  GHOSTCWT_MERGE_LEVELS=0 keeps every scale at its largest R.
  {
    if (option_or("merge_levels", 1) != 0) {
      std::map<int, std::vector<int>> by_r;
      for (int i = 0; i < prm.n_freqs; ++i)
        if (hp->scales[i].method == GCWT_SCALE_SPECTRAL) by_r[hp->scales[i].decimation].push_back(i);
      auto halo_of = [&](int r, const std::vector<int>& idx) {
        int h = 16;
        for (int i : idx) {
          int hs = std::max((int)std::ceil(hp->scales[i].support / (double)r) + halo_margin(), 16);
          if (r == 2) hs += hs & 1;
          h = std::max(h, hs);
        }
        return h;
      };
      const double kPrologue = 3.0;
      auto cost = [&](int r, const std::vector<int>& idx) {
        return (kPrologue + (double)idx.size()) / (double)(B - 2 * halo_of(r, idx));
      };
      std::vector<int> rs;
      for (const auto& kv : by_r) rs.push_back(kv.first);
      for (size_t k = rs.size(); k-- > 1;) {           // from the largest decimation down
        const int r = rs[k], r_lo = r / 2;
        if (r_lo < 2 || !by_r.count(r_lo) || by_r[r].empty() || by_r[r_lo].empty()) continue;
        std::vector<int> merged = by_r[r_lo];
        merged.insert(merged.end(), by_r[r].begin(), by_r[r].end());
        if (merged.size() > 256 || halo_of(r_lo, merged) > 32) continue;
        if (cost(r_lo, merged) < cost(r, by_r[r]) + cost(r_lo, by_r[r_lo])) {
          for (int i : by_r[r]) hp->scales[i].decimation = r_lo;
          by_r[r_lo] = merged;
          by_r[r].clear();
        }
      }
    }
  }
  plan_blockconv(hp, option_or("direct_max_len", -1) < 0 && !hp->exact_only);
  int64_t lmax_spec = longest_fft_kernel();
  for (int i = 0; i < prm.n_freqs; ++i) {
    ScalePlan& sp = hp->scales[i];
    if (sp.method == GCWT_SCALE_DIRECT) {
      sp.direct_index = hp->n_direct++;
      sp.direct_offset = hp->direct_total;
      hp->direct_total += ((sp.length + 7 + 7) & ~(int64_t)7) + 8;   // up to 7 zero taps in front, whole groups of 8
    } else if (sp.method == GCWT_SCALE_FULLBAND) {
      sp.fullband_index = hp->n_fullband++;
    }
  }

  // segments: one per epoch, or overlapping time blocks when an epoch needs a longer FFT
  int64_t pmin = INT64_MAX;
  for (int e = 0; e < n_ep; ++e) {
    const int64_t e0 = hp->bounds[2 * e], e1 = hp->bounds[2 * e + 1];
    auto add = [&](int64_t in0, int64_t in1, int64_t c0, int64_t c1, int64_t p, int64_t unit0 = -1, int64_t unit1 = -1) {
      EpochPlan ep;
      ep.start = in0 & ~(int64_t)63;            // aligned; samples before the epoch read as zero
      // [unit0, unit1): where the block's input keeps its full weight (time blocks with ramps); the rest fades out
      if (unit0 > in0) ep.ramp_lo = unit0 - ep.start;
      if (unit1 >= 0 && unit1 < in1) ep.ramp_hi = in1 - unit1;
      ep.lead = std::max<int64_t>(0, e0 - ep.start);
      ep.stop = in1; ep.ne = in1 - ep.start;
      ep.core0 = c0; ep.core1 = c1; ep.epoch = e;
      ep.p = p;
      ep.long_a = (int)std::max<int64_t>(1, p >> 22);
      ep.p_store = p / ep.long_a;
      ep.p1 = (int)(ep.p_store / kRowLen);
      pmin = std::min(pmin, p);
      hp->max_p = std::max(hp->max_p, p);
      hp->max_p_store = std::max(hp->max_p_store, ep.p_store);
      hp->epochs.push_back(ep);
    };
    const int64_t whole = std::max<int64_t>(kRowLen, next_pow2(e1 - (e0 & ~(int64_t)63) + lmax_spec));
    if (whole <= pmax) {
      add(e0, e1, e0, e1, whole);
      continue;
    }
    // time blocks: every output sample needs the input within (L-1)/2 of it; block
    // boundaries sit on multiples of 64 samples of the recording
    const int64_t halo_s = lmax_spec / 2 + 2;
    const int64_t ramp = hp->high_precision ? block_ramp(halo_s) : 0;       // EpochPlan::ramp_lo / ramp_hi
    const int64_t core_len = (pmax - 2 * halo_s - 2 * ramp - 64) & ~(int64_t)63;
    if (core_len < pmax / 4)
      return fail(GCWT_ERR_UNSUPPORTED,
                  "longest wavelet is too long for time blocks of 2^max_fft_log2 samples");
    for (int64_t c0 = e0; c0 < e1;) {
      const int64_t c1 = std::min(e1, (c0 + core_len) & ~(int64_t)63);
      // a side that would reach past the epoch's own end stops there: that edge is the recording's, and stays hard
      const int64_t u0 = c0 - halo_s, u1 = c1 + halo_s;
      const int64_t in0 = u0 - ramp > e0 ? u0 - ramp : e0, in1 = u1 + ramp < e1 ? u1 + ramp : e1;
      add(in0, in1, c0, c1, pmax, in0 > e0 ? u0 : -1, in1 < e1 ? u1 : -1);
      c0 = c1;
    }
  }
  const int n_seg = (int)hp->epochs.size();
  (void)n_seg;

  // Levels: one per decimation factor R in use.  GHOSTCWT_SPLIT_LEVELS=1 splits the scales of
  // an R by the block halo their kernels need into two levels that share x_R -- those that
  // fit the minimum halo of 16 (hop 224) and the rest (hop 212 for the default wavelet).
  // Measured on the headline workload: 14.84 ms against 14.30 ms unsplit (a workgroup's
  // prologue is paid per level), so it is off by default.
  const int r_cap = (int)std::min<int64_t>(kMaxDecimation, pmin / B);
  bool split = false;
  split = option_or("split_levels", 0) == 1;
  std::map<std::pair<int, int>, int> level_of;
  std::map<int, int> owner_of_r;
  for (int i = 0; i < prm.n_freqs; ++i) {
    ScalePlan& sp = hp->scales[i];
    if (sp.method != GCWT_SCALE_SPECTRAL) continue;
    const int r = sp.decimation;
    if (r > r_cap) return fail(GCWT_ERR_UNSUPPORTED, "internal: decimation beyond the shortest FFT");
    // Decimated samples discarded at each block edge: the kernel's measured support (the
    // reference length L is "4 footprints to be safe", morse.py:113-116; what lies beyond
    // `support` holds less than support_tol of the kernel's energy), plus two.
    int halo = std::max((int)std::ceil(sp.support / (double)r) + halo_margin(), 16);
    if (r == 2) halo += halo & 1;   // keeps halo*R a multiple of 4: 16-byte aligned tile runs
    // Heavy-tailed kernels are long against their band: within one decimation their block halos
    // range from about 40 to over 100 decimated samples.  Those above 48 form a level of their own
    // (sharing x_R): the rest keep their longer hop and the production kernel's static row layout.
    const std::pair<int, int> key(r, sp.theta_neg > 0.0 && halo > 48 ? 2 : (split && halo > 16 ? 1 : 0));
    auto it = level_of.find(key);
    if (it == level_of.end()) {
      LevelPlan lp;
      lp.decimation = r;
      const int idx = (int)hp->levels.size();
      if (!owner_of_r.count(r)) owner_of_r[r] = idx;
      lp.xr_owner = owner_of_r[r];
      level_of[key] = idx;
      hp->levels.push_back(lp);
      it = level_of.find(key);
    }
    sp.level = it->second;
    LevelPlan& lp = hp->levels[sp.level];
    lp.scales.push_back(i);
    lp.halo = std::max(lp.halo, halo);
  }
  // Band shift of every decimation (the levels of one decimation share x_R, so they share the
  // shift): the bins below zero frequency its scales need, in steps of max(1, R / 16) bins.  A
  // member whose upper edge no longer fits (cannot happen for bands that scale with the
  // frequency, as a wavelet family's do) would be a planning error.
  {
    std::map<int, int> shift_of_r;
    for (const LevelPlan& lp : hp->levels) {
      const double delta = 2.0 * M_PI / ((double)B * lp.decimation);
      int& shift = shift_of_r[lp.decimation];
      for (int i : lp.scales)
        if (hp->scales[i].theta_neg > 0.0)
          shift = std::max(shift, (int)std::ceil(hp->scales[i].theta_neg / delta));
    }
    for (LevelPlan& lp : hp->levels) {
      const double delta = 2.0 * M_PI / ((double)B * lp.decimation);
      const int g = std::max(1, lp.decimation / 16);
      const int shift = (shift_of_r[lp.decimation] + g - 1) / g * g;
      lp.band_shift = shift;
      for (int i : lp.scales)
        if (hp->scales[i].theta_hi > (double)(B - shift) * delta * (1.0 + 1e-12))
          return fail(GCWT_ERR_UNSUPPORTED, "internal: a scale's band does not fit its level's shifted band");
      if (shift > 0 && lp.scales.size() > 256)
        return fail(GCWT_ERR_UNSUPPORTED, "more than 256 heavy-tailed scales in one decimation level");
    }
  }
  // precision = high: the low cut of every x_R, below the band of all the scales that read it
  // (a decimation's levels share x_R; a shifted band holds the negative frequencies: no cut)
  if (hp->high_precision) {
    std::vector<double> lo(hp->levels.size(), 1e30);
    for (const LevelPlan& lp : hp->levels) {
      double& o = lo[(size_t)lp.xr_owner];
      if (lp.band_shift > 0) o = 0.0;
      for (int i : lp.scales) o = std::min(o, hp->scales[i].theta_lo);
    }
    for (size_t l = 0; l < hp->levels.size(); ++l)
      hp->levels[l].taper_hi = hp->levels[l].xr_owner == (int)l && lo[l] < 1e29 ? lo[l] : 0.0;
  }
  for (LevelPlan& lp : hp->levels) {
    lp.hop = B - 2 * lp.halo;
    if (lp.hop < 32)
      return fail(GCWT_ERR_UNSUPPORTED,
                  "internal: a spectral scale does not fit its 256-sample decimated block");
    lp.fast = lp.halo <= 48 && lp.scales.size() <= 256;                     // fast kernel's limits
    if (!lp.fast) hp->halo_static = false;
    lp.twiddle_offset = hp->level_twiddle_total;
    hp->level_twiddle_total += (int64_t)kSynthCols * lp.decimation;
  }
  // GHOSTCWT_INTERP=0: every level on the FFT-per-sample kernels (A/B runs, tests)
  {
    if (option_or("interp", 1) != 0) {
      // The levels' designs are independent (each writes its own LevelPlan and the demodulation bins of its own
      // scales) and are most of a first plan's host time (config 5: 120 ms): side by side on up to eight threads,
      // their tables appended in level order afterwards
      const size_t nl = hp->levels.size();
      std::vector<std::vector<float>> coef(nl);
      const unsigned n_thr = (unsigned)std::min<size_t>(nl, std::min<unsigned>((unsigned)std::max<long long>(1, option_or("plan_threads", 8)), std::max(1u, std::thread::hardware_concurrency())));
      if (n_thr <= 1) {
        for (size_t l = 0; l < nl; ++l) plan_interp_level(hp, &hp->levels[l], &coef[l]);
      } else {
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        auto work = [&]() {
          try {
            for (size_t l = next++; l < nl; l = next++) plan_interp_level(hp, &hp->levels[l], &coef[l]);
          } catch (...) {                  // (an exception must not leave a thread; the caller reports it)
            failed = true;
          }
        };
        std::vector<std::thread> pool;
        try {
          for (unsigned k = 1; k < n_thr; ++k) pool.emplace_back(work);
        } catch (...) {
          // no more threads to be had: this one and those that did start take the levels (nothing throws across the ABI)
        }
        work();
        for (std::thread& t : pool) t.join();
        if (failed) throw std::bad_alloc();
      }
      for (size_t l = 0; l < nl; ++l) {
        if (hp->levels[l].interp_q <= 0 || coef[l].empty()) continue;
        hp->levels[l].coef_offset = (int64_t)hp->interp_coef.size();
        hp->interp_coef.insert(hp->interp_coef.end(), coef[l].begin(), coef[l].end());
      }
    }
  }

  // long mode (FFT lengths 2^23, 2^24): what the combined low half of the spectrum can serve
  for (const EpochPlan& ep : hp->epochs) {
    if (ep.long_a == 1) continue;
    if (!hp->high_precision)
      return fail(GCWT_ERR_UNSUPPORTED, "FFT lengths above 2^22 (kernels of millions of taps) need precision = high");
    if (hp->n_fullband > 0)
      return fail(GCWT_ERR_UNSUPPORTED, "FFT lengths above 2^22 cannot serve full-band scales (heavy-tailed wavelet at very low frequencies)");
    for (const LevelPlan& lp : hp->levels) {
      const int64_t m = ep.p / lp.decimation;
      if (lp.band_shift > 0 || lp.decimation < 2 * ep.long_a ||
          (lp.decimation / ep.long_a > kMaxTwoPassDecimation && m > 8192))
        return fail(GCWT_ERR_UNSUPPORTED,
                    "FFT lengths above 2^22 need every decimation level at R >= 2 * (FFT length / 2^22): the plan mixes "
                    "kernels of millions of taps with scales near Nyquist (split the frequency list in two plans)");
    }
  }
  // per-segment block ranges of every level
  for (EpochPlan& ep : hp->epochs) {
    ep.lv.resize(hp->levels.size());
    for (size_t l = 0; l < hp->levels.size(); ++l) {
      const LevelPlan& lp = hp->levels[l];
      EpochLevel& el = ep.lv[l];
      el.m = ep.p / lp.decimation;
      // blocks whose kept samples [b*hop*R, (b+1)*hop*R) meet the output range (segment-local)
      const int64_t w_lo = ep.core0 - ep.start, w_hi = ep.core1 - ep.start;
      const int64_t span = (int64_t)lp.hop * lp.decimation;
      el.blk_lo = (int)(w_lo / span);
      el.nblk = (int)((w_hi + span - 1) / span) - el.blk_lo;
    }
  }
  // Batches: consecutive segments of one FFT length run as one launch set.  They share the
  // level grids (the union of their block ranges: a block past a segment's own range only
  // produces samples outside its window, which the stores drop) and each owns a slot of
  // the workspace, so the batch size is bounded by a memory budget.
  const int64_t C = prm.n_channels;
  int64_t budget = (int64_t)6 << 30;
  budget = std::max<int64_t>(0, option_or("batch_bytes", budget));
  for (size_t first = 0; first < hp->epochs.size();) {
    EpochPlan& lead = hp->epochs[first];
    int64_t blocks = 0;
    for (size_t l = 0; l < hp->levels.size(); ++l) blocks += (int64_t)lead.lv[l].nblk * B;
    // X + x_R (< P) + XB (+ Z), roughly; precision = high adds the float64 intermediate of the forward transform,
    // 16 bytes for rows 0 .. P1/2 of the spectrum (every row when full-band scales read all of it)
    const int64_t y_rows = !hp->high_precision ? 0 : (lead.p1 >= 4 && hp->n_fullband == 0 ? lead.p1 / 2 + 1 : lead.p1);
    const int64_t per_slot = 8 * C * (lead.p + lead.p + 2 * blocks + std::min(hp->n_fullband, 4) * lead.p) +
                             16 * C * y_rows * 4096;
    int cap = (int)std::max<int64_t>(1, std::min<int64_t>(kMaxBatch, budget / std::max<int64_t>(1, per_slot)));
    cap = (int)std::max<int64_t>(1, std::min<int64_t>(cap, 65535 / C));   // grid.y = segments * channels
    size_t count = 1;
    while (first + count < hp->epochs.size() && (int)count < cap && hp->epochs[first + count].p == lead.p)
      ++count;
    // shared grids
    std::vector<EpochLevel> lv(hp->levels.size());
    std::vector<SynthItem> items;
    int64_t xr = 0, xb = 0;
    for (size_t l = 0; l < hp->levels.size(); ++l) {
      const LevelPlan& lp = hp->levels[l];
      int lo = INT32_MAX, hi = 0;
      for (size_t i = 0; i < count; ++i) {
        const EpochLevel& el = hp->epochs[first + i].lv[l];
        lo = std::min(lo, el.blk_lo);
        hi = std::max(hi, el.blk_lo + el.nblk);
      }
      EpochLevel& g = lv[l];
      g.m = lead.p / lp.decimation;
      g.blk_lo = lo;
      g.nblk = hi - lo;
      g.xr_offset = lp.xr_owner == (int)l ? xr : lv[lp.xr_owner].xr_offset;   // x_R is per decimation
      g.xb_offset = xb;
      if (lp.xr_owner == (int)l) xr += g.m;
      xb += (int64_t)g.nblk * B;
      const int group = std::max(1, 64 / std::min(lp.decimation, 64));
      for (int b0 = 0; b0 < g.nblk; b0 += group)
        for (int sc : lp.scales)
          items.push_back(SynthItem{(int32_t)l, (int32_t)sc, (int32_t)b0,
                                    (int32_t)std::min(group, g.nblk - b0)});
    }
    for (size_t i = 0; i < count; ++i) {
      EpochPlan& ep = hp->epochs[first + i];
      ep.lv = lv;
      ep.items = items;
      ep.xr_total = xr;
      ep.xb_total = xb;
      ep.batch_first = (int)first;
      ep.batch_count = i == 0 ? (int)count : 0;
    }
    hp->max_xr = std::max(hp->max_xr, xr);
    hp->max_xb = std::max(hp->max_xb, xb);
    hp->max_batch = std::max(hp->max_batch, (int)count);
    first += count;
  }

  if (hp->n_blockconv > 0) {
    // block spectra of every channel for as many blocks as 1 GB holds (the launches walk the recording in such
    // chunks), never more than the finest group needs
    int min_hop = kRowLen;
    for (const HostPlan::BcGroup& g : hp->bc_groups) min_hop = std::min(min_hop, g.hop);
    const int64_t most = prm.n_samples / min_hop + 2 * (int64_t)n_ep + 1;
    hp->bc_chunk_blocks = std::max<int64_t>(2, std::min<int64_t>(most + 2, ((int64_t)1 << 30) / (8 * kRowLen * C)) & ~(int64_t)1);   // whole pairs
  }
  hp->workspace_bytes = 8 * C * hp->max_batch * (hp->max_p_store + hp->max_xr + hp->max_xb)   // X, x_R, XB
                        + 8 * kRowLen * (C * hp->bc_chunk_blocks + hp->n_blockconv)             // block spectra, responses
                        + 8 * std::min<int64_t>(hp->n_fullband, 4) * (C * hp->max_batch + 1) * hp->max_p  // Z, H
                        + 8 * (int64_t)hp->amps.size()
                        + 8 * (int64_t)prm.n_freqs * B                  // bank
                        + 8 * (hp->direct_total + hp->level_twiddle_total + kRowLen + 256)
                        + 16 * C;
  return GCWT_OK;
}

}  // namespace gcwt
