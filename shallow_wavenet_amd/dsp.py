"""Host-side signal processing either side of the network: run.sh stages 3 / 6 / 9, i.e. the reference's
`src/bin/noise_shaping.py` and the numpy / scipy helpers of `src/bin/feature_extract.py`.

The reference does this with pysptk (SPTK's MLSA filter) and pyworld - C libraries that are neither vendored by the reference nor
installed here.  What is restated in this package:
  * `mc2b`, `MLSAFilter.synthesis` - csrc/swn_dsp.c (plain C, libswn_dsp.so): the MLSA synthesis filter of
    noise_shaping.py:51-86 from its published algorithm, pinned by its transfer function (tests/test_dsp.py); parity with pysptk
    itself is unpinned;
  * `low_cut_filter`, `low_pass_filter`, `continuous_f0` - feature_extract.py:57-76,128-182 on scipy;
  * `noise_shaping` - noise_shaping.py:144-181 (the frame count WORLD's harvest would return is computed, not analysed);
  * `world_analysis` / `extract_features` - feature_extract.py:79-113,276-312: thin calls into pyworld / pysptk when they are importable,
    a clear ImportError otherwise (WORLD itself is not restated).
CPU code; nothing here is on the MI355X hot path."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DSP_PATH = os.path.join(_HERE, "libswn_dsp.so")
_dsp = None
_DP = ctypes.POINTER(ctypes.c_double)


def _lib():
    global _dsp
    if _dsp is None:
        if not os.path.exists(_DSP_PATH):
            raise RuntimeError(f"{_DSP_PATH} is missing: build it with `make -C shallow_wavenet_amd/csrc`")
        lib = ctypes.CDLL(_DSP_PATH)
        lib.swn_dsp_mc2b.argtypes = [_DP, ctypes.c_int, ctypes.c_double, _DP]
        lib.swn_dsp_mc2b.restype = None
        lib.swn_dsp_b2mc.argtypes = [_DP, ctypes.c_int, ctypes.c_double, _DP]
        lib.swn_dsp_b2mc.restype = None
        lib.swn_dsp_mlsa_synthesis.argtypes = [_DP, ctypes.c_long, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                               ctypes.c_int, _DP]
        lib.swn_dsp_mlsa_synthesis.restype = ctypes.c_int
        _dsp = lib
    return _dsp


def _rows(fn, a: np.ndarray, alpha: float) -> np.ndarray:
    a = np.ascontiguousarray(np.atleast_2d(a), dtype=np.float64)
    out = np.empty_like(a)
    for i in range(a.shape[0]):
        fn(a[i].ctypes.data_as(_DP), a.shape[1] - 1, float(alpha), out[i].ctypes.data_as(_DP))
    return out


def mc2b(mc, alpha: float) -> np.ndarray:
    """mel-cepstrum (.., M + 1) -> MLSA filter coefficients (pysptk.mc2b as noise_shaping.py:78 applies it per frame)"""
    mc = np.asarray(mc, dtype=np.float64)
    return _rows(_lib().swn_dsp_mc2b, mc, alpha).reshape(mc.shape)


def b2mc(b, alpha: float) -> np.ndarray:
    b = np.asarray(b, dtype=np.float64)
    return _rows(_lib().swn_dsp_b2mc, b, alpha).reshape(b.shape)


class MLSAFilter:
    """MLSA synthesis filter with per-frame coefficients interpolated inside every hop (pysptk.synthesis.Synthesizer around
    MLSADF(order, alpha), Pade order 4 as pysptk defaults to)."""

    def __init__(self, order: int, alpha: float, hopsize: int, pade: int = 4):
        if pade not in (4, 5):
            raise ValueError("Pade order must be 4 or 5")
        self.order, self.alpha, self.hopsize, self.pade = int(order), float(alpha), int(hopsize), int(pade)

    def synthesis(self, x, b) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        b = np.ascontiguousarray(np.atleast_2d(b), dtype=np.float64)
        if b.shape[1] != self.order + 1:
            raise ValueError(f"coefficient frames of {b.shape[1]} values for a filter of order {self.order}")
        if not np.isfinite(b).all():
            raise ValueError("non-finite filter coefficients")
        y = np.empty_like(x)
        rc = _lib().swn_dsp_mlsa_synthesis(x.ctypes.data_as(_DP), x.size, b.ctypes.data_as(_DP), b.shape[0], self.order, self.alpha,
                                           self.pade, self.hopsize, y.ctypes.data_as(_DP))
        if rc != 0:
            raise RuntimeError(f"swn_dsp_mlsa_synthesis failed ({rc})")
        return y


def synthesis_diff(x, diffmcep, alpha: float, fs: int, shiftms: float = 5.0) -> np.ndarray:
    """filter x with a differential mel-cepstrum sequence (T, dim + 1)   (noise_shaping.py:51-86 without the power modification,
    which its caller never requests)"""
    diffmcep = np.asarray(diffmcep, dtype=np.float64)
    b = mc2b(diffmcep, alpha)
    return MLSAFilter(diffmcep.shape[1] - 1, alpha, int(fs / 1000 * shiftms)).synthesis(x, b)


def world_frame_count(n_samples: int, fs: int, frame_period_ms: float = 5.0) -> int:
    """number of frames WORLD's analysers return for a signal of n_samples (the reference runs pw.harvest only to learn it)"""
    return int(1000.0 * n_samples / fs / frame_period_ms) + 1


def _fir(numtaps: int, cutoff: float, highpass: bool):
    from scipy.signal import firwin
    return firwin(numtaps, cutoff, pass_zero=not highpass)


def low_cut_filter(x, fs: int, cutoff: float = 70.0) -> np.ndarray:
    """255-tap FIR high-pass at `cutoff` Hz, causal (feature_extract.py:57-76: the group delay stays in the signal)"""
    from scipy.signal import lfilter
    return lfilter(_fir(255, cutoff / (fs // 2), True), 1, np.asarray(x, dtype=np.float64))


def low_pass_filter(x, fs: int, cutoff: float = 20.0) -> np.ndarray:
    """255-tap FIR low-pass with edge padding and the delay removed (feature_extract.py:128-150; smooths the F0 contour)"""
    from scipy.signal import lfilter
    taps = 255
    padded = np.pad(np.asarray(x, dtype=np.float64), (taps, taps), "edge")
    y = lfilter(_fir(taps, cutoff / (fs // 2), False), 1, padded)
    return y[taps + taps // 2: -taps // 2]


def continuous_f0(f0) -> Tuple[np.ndarray, np.ndarray]:
    """(voiced flags, F0 with the unvoiced stretches bridged linearly and both ends held)   (feature_extract.py:153-182)"""
    f0 = np.array(f0, dtype=np.float64)
    uv = (f0 != 0).astype(np.float32)
    voiced = np.flatnonzero(f0)
    if voiced.size == 0:
        raise ValueError("no voiced frame")
    return uv, np.interp(np.arange(f0.size), voiced, f0[voiced])


def noise_shaping(x, mean_mcep, fs: int, alpha: float, mag: float = 0.5, mcep_dim_start: int = 5, inv: bool = False,
                  shiftms: float = 5.0, cutoff: float = 70.0) -> np.ndarray:
    """one utterance of noise_shaping.py:144-181: the time-invariant MLSA filter built from the corpus-mean mel-cepstrum (the
    statistics vector from `mcep_dim_start` on, scaled by `mag`, c(0) zeroed; `inv` flips the sign of c(1..): run.sh APPLIES the shaping
    with `--inv true` before training (run.sh:529-543) and RESTORES decoded waveforms with `--inv false` (run.sh:725-740)), then the
    70 Hz low cut."""
    x = np.asarray(x, dtype=np.float64)
    coef = np.array(mean_mcep, dtype=np.float64)[mcep_dim_start:] * mag
    coef[0] = 0.0
    if inv:
        coef[1:] = -coef[1:]
    frames = np.tile(coef, (world_frame_count(x.size, fs, shiftms), 1))
    return low_cut_filter(synthesis_diff(x, frames, alpha, fs, shiftms), fs, cutoff)


def _need(mod: str):
    import importlib
    try:
        return importlib.import_module(mod)
    except ImportError as e:                                   # pragma: no cover - depends on the environment
        raise ImportError(f"{mod} is required for WORLD / SPTK analysis (run.sh stage 1); it is not part of this package and the "
                          f"reference does not vendor it") from e


def world_analysis(x, fs: int, minf0: float = 40.0, maxf0: float = 700.0, shiftms: float = 5.0, fftl: int = 1024):
    """harvest + cheaptrick + d4c through pyworld (feature_extract.py:95-113); ImportError when pyworld is absent"""
    pw = _need("pyworld")
    x = np.asarray(x, dtype=np.float64)
    f0, t = pw.harvest(x, fs, f0_floor=minf0, f0_ceil=maxf0, frame_period=shiftms)
    sp = pw.cheaptrick(x, f0, t, fs, fft_size=fftl)
    ap = pw.d4c(x, f0, t, fs, fft_size=fftl)
    return t, f0, sp, ap


def extract_features(x, fs: int, mcep_dim: int = 49, mcep_alpha: float = 0.455, **kw) -> np.ndarray:
    """the (T, 2 + n_codeap + mcep_dim + 1) matrix `[uv, log F0 (continuous, smoothed), coded aperiodicity, mel-cepstrum]` the
    network is conditioned on (feature_extract.py:296-312); needs pyworld and pysptk"""
    pw, ps = _need("pyworld"), _need("pysptk")
    _, f0, sp, ap = world_analysis(x, fs, **kw)
    uv, cf0 = continuous_f0(f0)
    shiftms = kw.get("shiftms", 5.0)
    lf0 = np.log(low_pass_filter(cf0, int(1.0 / (shiftms * 0.001)), cutoff=20))
    return np.concatenate([uv[:, None], lf0[:, None], pw.code_aperiodicity(ap, fs), ps.sp2mc(sp, mcep_dim, mcep_alpha)], axis=1)
