"""CPU: the host-side signal processing of run.sh stages 3 / 6 / 9 (shallow_wavenet_amd/dsp.py, csrc/swn_dsp.c).

pysptk / pyworld are absent (and not vendored by the reference), so parity with them is UNPINNED; the MLSA filter is pinned by the
published definition of what it realises instead:
    H(z) = exp( sum_m c(m) ((z^-1 - a) / (1 - a z^-1))^m )
to the accuracy of its Pade approximation (order 4: ~3e-5 relative for |log H| < 0.3, 1e-2 up to ~3), by b -> c -> b round trips, by
shaping followed by inverse shaping, and by linearity / time invariance.  The scipy helpers are checked on known signals."""
import os
import subprocess
import sys

import numpy as np
import pytest

from shallow_wavenet_amd import dsp, featio
from shallow_wavenet_amd.decode_driver import write_wav_pcm16
from shallow_wavenet_amd.noise_shaping_driver import main as ns_main, read_wav_fs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _definition(mc, alpha, n):
    w = np.linspace(0, np.pi, n // 2 + 1)
    z1 = np.exp(-1j * w)
    ap = (z1 - alpha) / (1 - alpha * z1)
    return np.exp(sum(mc[m] * ap ** m for m in range(len(mc))))


@pytest.mark.parametrize("order,alpha,scale,pade,tol", [(24, 0.455, 0.25, 4, 1e-4), (49, 0.466, 0.25, 4, 1e-4), (24, 0.41, 2.0, 4, 2e-2),
                                                        (24, 0.41, 2.0, 5, 5e-3), (4, 0.0, 0.5, 4, 1e-4)])
def test_mlsa_filter_realises_its_transfer_function(order, alpha, scale, pade, tol):
    rng = np.random.default_rng(order)
    mc = rng.normal(size=order + 1) * scale / (1 + np.arange(order + 1))
    b = dsp.mc2b(mc, alpha)
    assert np.allclose(dsp.b2mc(b, alpha), mc, atol=1e-14)
    n = 8192
    x = np.zeros(n); x[0] = 1.0
    h = dsp.MLSAFilter(order, alpha, hopsize=110, pade=pade).synthesis(x, b[None, :])
    want = _definition(mc, alpha, n)
    got = np.fft.rfft(h)
    assert np.max(np.abs(got - want) / np.abs(want)) < tol


def test_inverse_filter_restores_the_signal_and_the_filter_is_linear():
    rng = np.random.default_rng(3)
    order, alpha = 30, 0.455
    mc = rng.normal(size=order + 1) * 0.3 / (1 + np.arange(order + 1)); mc[0] = 0.0
    b = dsp.mc2b(np.tile(mc, (40, 1)), alpha)
    f = dsp.MLSAFilter(order, alpha, hopsize=110)
    x = rng.uniform(-0.5, 0.5, 4000)
    y = f.synthesis(x, b)
    assert np.max(np.abs(f.synthesis(y, -b) - x)) < 1e-10          # P(F) / P(-F) and its mirror image cancel exactly
    x2 = rng.uniform(-0.5, 0.5, 4000)
    assert np.allclose(f.synthesis(2.0 * x + x2, b), 2.0 * y + f.synthesis(x2, b), atol=1e-12)
    shifted = f.synthesis(np.concatenate([np.zeros(17), x])[:4000], b)    # constant coefficients: time invariant
    assert np.allclose(shifted[17:], y[:4000 - 17], atol=1e-12)


def test_coefficients_are_interpolated_inside_a_hop():
    order, alpha, hop = 8, 0.3, 50
    b = np.zeros((3, order + 1)); b[1, 0] = np.log(2.0)             # gain ramps from 1 to 2 over the first hop, back over the second
    y = dsp.MLSAFilter(order, alpha, hop).synthesis(np.ones(150), b)
    assert np.allclose(y[:50], np.exp(np.log(2.0) * np.arange(50) / 50))
    assert np.isclose(y[50], 2.0) and np.allclose(y[100:], 1.0)


def test_scipy_helpers():
    fs = 22050
    t = np.arange(fs) / fs
    x = 0.3 + 0.2 * np.sin(2 * np.pi * 1000 * t)
    y = dsp.low_cut_filter(x, fs, 70)                                # 255 taps resolve ~86 Hz: at 70 Hz the cut is a shelf, not a notch
    assert 0.0 < np.mean(y[2000:]) < 0.5 * 0.3                       # DC attenuated ...
    assert abs(np.std(y[2000:] - np.mean(y[2000:])) - 0.2 / np.sqrt(2)) < 2e-3   # ... 1 kHz untouched
    y5 = dsp.low_cut_filter(x, fs, 500)
    assert abs(np.mean(y5[2000:])) < 1e-3                            # a cutoff the filter can resolve removes it
    from scipy.signal import firwin, lfilter
    assert np.array_equal(y, lfilter(firwin(255, 70 / (fs // 2), pass_zero=False), 1, x))    # the reference's own two scipy calls
    f0 = np.array([0, 0, 100, 110, 0, 0, 130, 0], dtype=np.float64)
    uv, c = dsp.continuous_f0(f0)
    assert uv.tolist() == [0, 0, 1, 1, 0, 0, 1, 0]
    assert np.allclose(c, [100, 100, 100, 110, 110 + 20 / 3, 110 + 40 / 3, 130, 130])
    slow = dsp.low_pass_filter(np.concatenate([np.full(300, 5.0), np.full(300, 6.0)]), 200, cutoff=20)
    assert slow.shape == (600,) and abs(slow[10] - 5.0) < 1e-6 and abs(slow[-10] - 6.0) < 1e-6
    assert dsp.world_frame_count(22050, 22050) == 201


def test_noise_shaping_cli_and_its_inverse(tmp_path):
    """the stage as run.sh calls it: waveform directory + statistics file -> shaped wavs; `--inv 1` on the result gives the input back
    (up to the two 70 Hz low cuts and 16-bit PCM)"""
    fs = 22050
    rng = np.random.default_rng(0)
    t = np.arange(fs // 2) / fs
    x = 0.25 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.standard_normal(t.size)
    wdir, sdir, rdir = tmp_path / "wav", tmp_path / "ns", tmp_path / "restored"
    wdir.mkdir()
    write_wav_pcm16(str(wdir / "a.wav"), x, fs)
    mean = np.concatenate([[0.9, 5.0, -3.0, -4.0, 0.1], [1.5, 1.2, -0.6, 0.3, -0.2, 0.1], np.zeros(44)])   # [uv, lf0, codeap.., mcep 0..49]
    stats = str(tmp_path / "stats.npz")
    featio.write_stats(stats, "/feat_org_lf0", mean, np.ones_like(mean))
    common = ["--stats", stats, "--fs", str(fs), "--mcep_alpha", "0.455", "--mcep_dim_start", "5", "--mag", "0.5", "--verbose", "0"]
    assert ns_main(["--waveforms", str(wdir), "--writedir", str(sdir)] + common) == 0
    shaped, fs2 = read_wav_fs(str(sdir / "a.wav"))
    assert fs2 == fs and shaped.shape == x.shape and np.max(np.abs(shaped - x)) > 1e-2          # it did something
    want = dsp.noise_shaping(read_wav_fs(str(wdir / "a.wav"))[0], mean, fs, 0.455)
    assert np.max(np.abs(shaped - np.clip(want, -1, 1))) <= 1.0 / 32767 + 1e-9
    assert ns_main(["--waveforms", str(sdir), "--writedir", str(rdir), "--inv", "1"] + common) == 0
    back, _ = read_wav_fs(str(rdir / "a.wav"))
    d = 254                                                           # two 255-tap linear-phase low cuts: 2 x 127 samples of delay
    ref = x[: x.size - d]
    err = back[d:] - ref
    assert np.sqrt(np.mean(err[2000:] ** 2)) < 0.02 * np.sqrt(np.mean(ref[2000:] ** 2)) + 1e-3
    # wrong sampling frequency is an error, as in the reference
    assert ns_main(["--waveforms", str(wdir), "--writedir", str(sdir), "--stats", stats, "--fs", "16000", "--verbose", "0"]) == 1


def test_bin_shim_prints_help():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "shallow_wavenet_amd", "bin", "noise_shaping.py"), "--help"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--mcep_alpha" in r.stdout and "--inv" in r.stdout


def test_world_analysis_needs_pyworld():
    try:
        import pyworld  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="pyworld"):
            dsp.world_analysis(np.zeros(2205), 22050)
