// planner.cpp -- see planner.h
#include "planner.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cmath>
#include <map>

namespace gcwt {

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
  if (!(prm.fs > 0)) return fail(GCWT_ERR_INVALID, "Sampling rate must be positive");
  if (!(prm.gamma > 0)) return fail(GCWT_ERR_INVALID, "gamma must be positive");
  if (!(prm.beta > 0)) return fail(GCWT_ERR_INVALID, "beta must be positive");
  if (prm.out_mode < 0 || prm.out_mode > 2) return fail(GCWT_ERR_INVALID, "bad out_mode");
  if (prm.block != 0 && prm.block != 256)
    return fail(GCWT_ERR_UNSUPPORTED, "only block = 256 is built");

  hp->prm = prm;
  hp->block = 256;
  hp->band_eps = prm.band_eps > 0 ? prm.band_eps : 1e-9;
  if (hp->band_eps > 1e-3) return fail(GCWT_ERR_INVALID, "band_eps too large");
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

  const double g = prm.gamma, b = prm.beta;
  hp->w0 = std::exp((std::log(b) - std::log(g)) / g);                 // morseutils.py:315
  hp->base_length = (2.0 * std::sqrt(2.0) * std::sqrt(g * b)) / hp->w0 * 4.0;  // morse.py:115
  band_edges(g, b, hp->band_eps, &hp->u_lo, &hp->u_hi);

  const int B = hp->block;
  hp->scales.resize(prm.n_freqs);
  int64_t lmax_spec = 1;
  for (int i = 0; i < prm.n_freqs; ++i) {
    ScalePlan& sp = hp->scales[i];
    sp.freq_hz = hp->freqs[i];
    if (!(sp.freq_hz > 0)) return fail(GCWT_ERR_INVALID, "analysis frequencies must be positive");
    sp.omega = sp.freq_hz / (prm.fs / 2.0) * M_PI;                    // transforms.py:408-410
    sp.length = (int64_t)std::ceil(hp->w0 / sp.omega * hp->base_length);  // morse.py:118-122
    if (sp.length < 1) sp.length = 1;
    sp.half_delay = (double)(sp.length - 1) / 2.0 - (double)((sp.length - 1) / 2);
    if (hp->u_hi * sp.omega > M_PI) {
      sp.method = GCWT_SCALE_DIRECT;
      sp.direct_index = hp->n_direct++;
      sp.direct_offset = hp->direct_total;
      hp->direct_total += sp.length;
      if (sp.length > 65536)
        return fail(GCWT_ERR_UNSUPPORTED, "a scale whose filter reaches Nyquist has a kernel "
                                          "longer than 65536 taps");
    } else {
      sp.method = GCWT_SCALE_SPECTRAL;
      lmax_spec = std::max(lmax_spec, sp.length);
    }
  }

  // segments: one per epoch, or overlapping time blocks when an epoch needs a longer FFT
  hp->max_fft_log2 = prm.max_fft_log2 == 0 ? 22 : prm.max_fft_log2;
  if (hp->max_fft_log2 < 12 || hp->max_fft_log2 > 22)
    return fail(GCWT_ERR_INVALID, "max_fft_log2 must be 0 or 12..22");
  const int64_t pmax = (int64_t)1 << hp->max_fft_log2;
  int64_t pmin = INT64_MAX;
  for (int e = 0; e < n_ep; ++e) {
    const int64_t e0 = hp->bounds[2 * e], e1 = hp->bounds[2 * e + 1];
    auto add = [&](int64_t in0, int64_t in1, int64_t c0, int64_t c1, int64_t p) {
      EpochPlan ep;
      ep.start = in0 & ~(int64_t)63;            // aligned; samples before the epoch read as zero
      ep.lead = std::max<int64_t>(0, e0 - ep.start);
      ep.stop = in1; ep.ne = in1 - ep.start;
      ep.core0 = c0; ep.core1 = c1; ep.epoch = e;
      ep.p = p; ep.p1 = (int)(p / kRowLen);
      pmin = std::min(pmin, p);
      hp->max_p = std::max(hp->max_p, p);
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
    const int64_t core_len = (pmax - 2 * halo_s - 64) & ~(int64_t)63;
    if (core_len < pmax / 4)
      return fail(GCWT_ERR_UNSUPPORTED,
                  "longest wavelet is too long for time blocks of 2^max_fft_log2 samples");
    for (int64_t c0 = e0; c0 < e1;) {
      const int64_t c1 = std::min(e1, (c0 + core_len) & ~(int64_t)63);
      add(std::max(e0, c0 - halo_s), std::min(e1, c1 + halo_s), c0, c1, pmax);
      c0 = c1;
    }
  }
  const int n_seg = (int)hp->epochs.size();
  (void)n_seg;

  // decimation factor per spectral scale, levels
  const int r_cap = (int)std::min<int64_t>(kMaxDecimation, pmin / B);
  std::map<int, int> level_of_r;
  for (int i = 0; i < prm.n_freqs; ++i) {
    ScalePlan& sp = hp->scales[i];
    if (sp.method != GCWT_SCALE_SPECTRAL) continue;
    int r = 2;  // u_hi*omega <= pi holds here, so R = 2 always fits
    while (2 * r <= r_cap && hp->u_hi * sp.omega * (2.0 * r) <= 2.0 * M_PI) r *= 2;
    sp.decimation = r;
    auto it = level_of_r.find(r);
    if (it == level_of_r.end()) {
      LevelPlan lp;
      lp.decimation = r;
      level_of_r[r] = (int)hp->levels.size();
      hp->levels.push_back(lp);
      it = level_of_r.find(r);
    }
    sp.level = it->second;
    LevelPlan& lp = hp->levels[sp.level];
    lp.scales.push_back(i);
    // decimated samples discarded at each block edge: half the kernel's effective support.
    // The reference length L is "4 footprints to be safe" (morse.py:113-116); beyond
    // 0.82 L/2 the wavelet is below 2e-5 of its peak and the wrap-around error of the
    // block convolution stays below 3e-7 of the row maximum (tests/test_host_surface.py).
    int halo = (int)std::ceil(hp->halo_frac * (double)sp.length / (2.0 * r)) + 2;
    halo = std::max(halo, 16);
    if (r == 2) halo += halo & 1;   // keeps halo*R a multiple of 4: 16-byte aligned tile runs
    lp.halo = std::max(lp.halo, halo);
  }
  for (LevelPlan& lp : hp->levels) {
    lp.hop = B - 2 * lp.halo;
    if (lp.hop < 32)
      return fail(GCWT_ERR_UNSUPPORTED,
                  "a wavelet is too long for the 256-sample decimated block at the largest "
                  "decimation this build supports (256): lowest frequency too low for fs");
    if (lp.halo > 32 || lp.scales.size() > 256) hp->halo_static = false;   // fast kernel's limits
    lp.twiddle_offset = hp->level_twiddle_total;
    hp->level_twiddle_total += (int64_t)kSynthCols * lp.decimation;
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
  if (const char* e = getenv("GHOSTCWT_BATCH_BYTES")) budget = std::max<int64_t>(0, atoll(e));
  for (size_t first = 0; first < hp->epochs.size();) {
    EpochPlan& lead = hp->epochs[first];
    int64_t blocks = 0;
    for (size_t l = 0; l < hp->levels.size(); ++l) blocks += (int64_t)lead.lv[l].nblk * B;
    const int64_t per_slot = 8 * C * (lead.p + lead.p + 2 * blocks);   // X + x_R (< P) + XB, roughly
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
      g.xr_offset = xr;
      g.xb_offset = xb;
      xr += g.m;
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

  hp->workspace_bytes = 8 * C * hp->max_batch * (hp->max_p + hp->max_xr + hp->max_xb)   // X, x_R, XB
                        + 8 * (int64_t)prm.n_freqs * B                  // bank
                        + 8 * (hp->direct_total + hp->level_twiddle_total + kRowLen + 256)
                        + 16 * C;
  return GCWT_OK;
}

}  // namespace gcwt
