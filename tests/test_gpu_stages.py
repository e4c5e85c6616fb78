"""Stage-by-stage parity of the HIP pipeline against the NumPy model (GPU)."""
import numpy as np
import pytest

from decimated_model import low_cut
from scipy.fft import fft, ifft

from conftest import rel_err
from oracle import ghost_oracle as orc

pytestmark = pytest.mark.gpu


def _plan_and_run(x, fs, freqs, **kw):
    from ghost_amd.engine import CwtPlan
    x = np.atleast_2d(np.asarray(x, dtype=np.float32))
    plan = CwtPlan(x.shape[1], x.shape[0], fs, freqs, **kw)
    out = plan.execute(x)
    return plan, out


def test_filter_bank_is_the_response_of_the_reference_kernel():
    """The device bank against the DTFT of the literal kernel (oracle), default and other
    Morse parameters; and, for the default one, against the continuous spectrum it is
    indistinguishable from."""
    from ghost_amd.engine import CwtPlan
    fs = 1000.0
    f = np.geomspace(200, 2, 100)
    k = np.arange(256)
    for gamma, beta in ((3, 20), (3, 8), (4, 30), (6, 100)):
        plan = CwtPlan(1 << 16, 1, fs, f, gamma=gamma, beta=beta)
        bank = plan.filter_bank()
        si = plan.scale_info()
        assert np.all(si["method"] == 0)
        om = orc.hz_to_rad(f, fs)
        for i in range(0, 100, 9):
            theta = 2 * np.pi * k / (256 * si["decimation"][i])
            ref = orc.kernel_response(theta, om[i], si["length"][i], gamma, beta)
            assert np.abs(bank[i] - ref).max() < 2e-7 * 2.0, (gamma, beta, i)
            if (gamma, beta) == (3, 20):
                closed = orc.spectral_filter(theta, om[i], si["length"][i])
                assert np.abs(bank[i] - closed).max() < 3e-7 * 2.0


def test_direct_kernel_matches_reference_psi(golden):
    from ghost_amd.engine import CwtPlan
    g = golden("g3_kernels.npz")
    fs = 1000.0
    plan = CwtPlan(4096, 1, fs, [391.0, 350.0])
    assert list(plan.scale_info()["method"]) == [1, 1]
    for i, L in enumerate((36, 40)):
        psi = plan.direct_kernel(i)
        ref = g["psi_%d" % L]
        assert psi.shape == ref.shape
        assert np.abs(psi - ref).max() < 2e-7 * np.abs(ref).max()


def test_forward_decimate_block_stages(golden, option):
    # the production kernel makes the block spectra in its own prologue: keep the separate
    # pass here so that the XB array exists to be looked at
    option("fuse_blocks", 0)
    g = golden("g1_config1.npz")
    x = g["x"]
    fs = float(g["fs"])
    plan, out = _plan_and_run(x, fs, g["frequencies"], output="complex")
    P = plan.info["fft_length"]
    p1 = P // 4096
    xc = x.astype(np.float64) - x.astype(np.float64).mean()
    X = fft(xc, n=P)
    # spectrum is stored k1-major: X~[k1*4096 + k2] = X[k1 + P1*k2]
    # only k2 < 2048 (X[k < P/2], the positive frequencies) is stored: nothing reads the rest
    got = plan.debug_fetch(0).astype(np.complex128).reshape(p1, 4096)[:, :2048]
    ref = X.reshape(4096, p1).T[:, :2048]
    assert np.abs(got - ref).max() < 2e-6 * np.abs(ref).max()
    for l, lv in enumerate(plan.debug_levels()):
        R, M = lv["decimation"], lv["m"]
        # (precision = high: the level's slice is cut below the band of its scales before the inverse)
        xr_ref = ifft(X[:M] * low_cut(lv["low_cut"], P, M)) * M            # unnormalised inverse = P * x_lp[R m]
        xr = plan.debug_fetch(1, level=l).astype(np.complex128)
        assert np.abs(xr - xr_ref).max() < 3e-6 * np.abs(xr_ref).max(), (R,)
        xb = plan.debug_fetch(2, level=l).astype(np.complex128).reshape(lv["nblk"], 256)
        for b in (0, lv["nblk"] - 1):
            idx = (b * lv["hop"] - lv["halo"] + np.arange(256)) % M
            ref_b = fft(xr_ref[idx]) / (256.0 * P)
            assert np.abs(xb[b] - ref_b).max() < 3e-6 * np.abs(ref_b).max(), (R, b)


def test_config1_complex_vs_golden(golden):
    g = golden("g1_config1.npz")
    plan, out = _plan_and_run(g["x"], float(g["fs"]), g["frequencies"], output="complex")
    err = rel_err(out[0][:, g["cols"]], g["complex_cols"])
    print("config1 complex rel err per scale:", err)
    assert err.max() < 1e-5


@pytest.mark.parametrize("n, f_hi, f_lo, gamma_beta", [(3000, 200.0, 20.0, (3, 20)), (16384, 200.0, 5.0, (3, 20)),
                                                       (300000, 200.0, 2.0, (3, 20)), (1000000, 200.0, 2.0, (3, 20)),
                                                       (2000000, 200.0, 1.0, (3, 20)), (1000000, 100.0, 0.1, (3, 4))])
def test_band_energies_of_the_detector_are_those_of_the_spectrum(n, f_hi, f_lo, gamma_beta):
    """precision = 'auto' predicts from the spectrum's band energies, sixteen bands per octave of the bin index
    (kernels.h: spec_band).  They are summed inside the forward row pass (fwd64.hip: row_band_sums -- wave butterflies
    over aligned runs of k2, the reflected rows from their twins) and added up row by row (detect.hip: k_band_sums):
    here against |FFT|^2 of the same recording, for FFT lengths 2^12 .. 2^21 -- one row, radix-2 columns, two real
    columns per transform, subsequences -- and for a plan with a full-band scale beside the decimated ones
    (Morse(3, 4) down to 0.1 Hz), which makes every row of the spectrum itself instead of reflecting half of them.
    The recording carries a mains line 40 x its spread: one bin that must land in its band and nowhere else."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd.synthetic import lfp
    fs = 1000.0
    x = lfp(1, n)[0].astype(np.float64)
    x += 40.0 * x.std() * np.sin(2 * np.pi * 60.0 * np.arange(n) / fs)
    x = x.astype(np.float32)
    f = np.geomspace(f_hi, f_lo, 24)
    plan = CwtPlan(n, 1, fs, f, gamma=gamma_beta[0], beta=gamma_beta[1], precision="auto")
    if gamma_beta != (3, 20):
        assert set(plan.scale_info()["method"].tolist()) >= {0, 2}           # decimated and full-band scales
    plan.execute(x[None, :])
    got = plan.debug_precision_terms()["band_energy"].astype(np.float64)
    P = plan.info["fft_length"]
    xc = x.astype(np.float64) - x.astype(np.float64).mean()
    E = np.abs(fft(xc, n=P)[1:P // 2]) ** 2
    band = (np.arange(1, P // 2, dtype=np.float32).view(np.uint32) >> 19).astype(np.int64) - 127 * 16
    ref = np.bincount(band, weights=E, minlength=384)[:384]
    assert got.shape == (384,)
    assert np.all(got[ref == 0] == 0)
    assert np.abs(got - ref).max() < 1e-5 * ref.max()
    big = ref > 1e-6 * ref.max()
    assert np.abs(got[big] / ref[big] - 1).max() < 2e-5, np.abs(got[big] / ref[big] - 1).max()
    # twice the same bits: fixed order of every sum
    plan.execute(x[None, :])
    again = plan.debug_precision_terms()["band_energy"]
    assert np.array_equal(again.view(np.uint32), got.astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("n, offset", [(1000000, 0.0), (700001, 1000.0), (4096 * 200, -37.5), (524288 + 5000, 1.0e4),
                                       (1500001, 300.0), (3000000, -1000.0), (300000, 50.0), (20000, 1000.0), (3000, 10.0)])
def test_channel_means_taken_inside_the_forward_passes(option, n, offset):
    """transforms.py:142-143 subtracts the recording's mean before anything else.  For a plan that is one segment of
    the whole recording (every scale spectral; FFTs of 2^12 .. 2^22 points: every variant of the column pass -- one
    row, radix-2 columns, two real columns per transform, two or four subsequences) the forward column pass sums the samples it reads
    and the row pass takes the mean's transform out of its input in float64 (fwd64.hip; option fold_mean = 0: a pass
    of its own over x first): per bin the spectrum is that of x - mean to float32 rounding, with an offset of ten
    thousand times the signal's spread as without one (float64: the offset costs 1e-16 of itself), the rows are the
    oracle's, and a block request gives the bits of the whole transform."""
    from ghost_amd.engine import CwtPlan
    from ghost_amd.synthetic import lfp
    fs = 1000.0
    x = (lfp(3, n).astype(np.float64) + offset).astype(np.float32)
    x[1] *= 0.25
    x[2] = x[2] - offset                                       # one channel without the offset
    f = np.geomspace(200.0, 2.0 if n >= 300000 else (10.0 if n >= 20000 else 40.0), 24 if n <= 1000000 else 8)
    plan = CwtPlan(n, 3, fs, f, output="complex")
    assert np.all(plan.scale_info()["method"] == 0)
    assert plan.info["fft_length"] == {1000000: 1 << 20, 700001: 1 << 20, 4096 * 200: 1 << 20, 524288 + 5000: 1 << 20,
                                       1500001: 1 << 21, 3000000: 1 << 22, 300000: 1 << 19}.get(n, plan.info["fft_length"])
    got = plan.execute(x)
    assert plan.debug_mean_folded()
    P, p1 = plan.info["fft_length"], plan.info["fft_length"] // 4096
    for c in range(3):
        xc = x[c].astype(np.float64) - x[c].astype(np.float64).mean()
        ref = fft(xc, n=P).reshape(4096, p1).T[:, :2048]
        spec = plan.debug_fetch(0, channel=c).reshape(p1, 4096)[:, :2048].astype(np.complex128)   # (the rest is never written)
        # per bin: float32 rounding of the bin itself plus 1e-9 of the spectrum's largest (float64 arithmetic on a
        # recording whose offset is up to 1e4 of its spread)
        assert np.all(np.abs(spec - ref) <= 1.5e-7 * np.abs(ref) + 1e-9 * np.abs(ref).max()), c
    b0, bl = (123457, 50001) if n > 200000 else (n // 7, n // 3)
    blk = plan.execute_block(x, b0, bl)
    np.testing.assert_array_equal(blk, got[:, :, b0:b0 + bl])
    ref = orc.cwt_complex(x[1].astype(np.float64), fs, f, n_threads=8)
    assert rel_err(got[1], ref).max() < 1e-5
    option("fold_mean", 0)
    plain = CwtPlan(n, 3, fs, f, output="complex")
    old = plain.execute(x)
    assert not plain.debug_mean_folded()
    assert rel_err(got.reshape(-1, n), old.reshape(-1, n)).max() < 2e-6
    # epochs cut out of the recording: the sums by their own pass, as before
    option("fold_mean", None)
    cut = CwtPlan(n, 3, fs, f, epoch_bounds=[[0, n // 2], [n // 2 + 7, n]])
    cut.execute(x)
    assert not cut.debug_mean_folded()
