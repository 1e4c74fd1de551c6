"""Deterministic synthetic parameters and inputs (no checkpoints exist offline).

SURVEY.md section 7.1 / 8(d): weights come from a *formula*, never from stored files, so
fixtures only carry inputs, noise and expected outputs.  PCG64 streams are stable across
numpy versions, so the GPU box regenerates bit-identical tensors.

Two weight sets:
  * "xavier"  - Xavier-uniform bound per tensor (what reference `initialize` draws from,
                cswnv_shift1.py:20-34) but with small non-zero biases so every bias path
                is exercised;
  * "trained" - same, plus out_2 scale rows biased so the Laplace scale b ~ 1e-2 and the
                mean rows scaled down: samples no longer saturate at +-1, which would hide
                errors (SURVEY.md section 7.3 "fp32 parity under feedback").
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

from .config import NetConfig


def _fans(shape):
    if len(shape) < 2:
        return 1, 1
    rf = 1
    for s in shape[2:]:
        rf *= s
    return shape[1] * rf, shape[0] * rf


def synth_state_dict(cfg: NetConfig, seed: int = 1234, flavor: str = "xavier",
                     identity_scale_in: bool = False) -> Dict[str, np.ndarray]:
    """fp32 numpy state dict with the reference's keys/shapes."""
    assert flavor in ("xavier", "trained")
    out: Dict[str, np.ndarray] = {}
    for ti, (name, shape) in enumerate(cfg.param_shapes()):
        rng = np.random.Generator(np.random.PCG64([seed, ti]))
        if name.endswith(".bias"):
            w = rng.uniform(-0.05, 0.05, size=shape)
        else:
            fan_in, fan_out = _fans(shape)
            bound = math.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-bound, bound, size=shape)
        if name == "upsampling.conv.weight":
            w = rng.uniform(0.75, 1.25, size=shape)
        if name == "upsampling.conv.bias":
            w = rng.uniform(-0.02, 0.02, size=shape)
        if identity_scale_in and name == "scale_in.weight":
            w = np.eye(shape[0]).reshape(shape)
        if identity_scale_in and name == "scale_in.bias":
            w = np.zeros(shape)
        if flavor == "trained" and cfg.kind == "laplace":
            if name == "out_2.weight":
                w = w.copy()
                w[: cfg.seg] *= 0.5            # mean rows
                w[cfg.seg: 2 * cfg.seg] *= 0.25    # scale rows
                w[2 * cfg.seg:] *= 0.25            # LP rows
            if name == "out_2.bias":
                w = w.copy()
                w[cfg.seg: 2 * cfg.seg] = -4.6 + w[cfg.seg: 2 * cfg.seg]   # sigmoid(-4.6) ~ 1e-2
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def synth_aux(cfg: NetConfig, batch: int, n_frames: int, seed: int = 1) -> np.ndarray:
    """aux ~ N(0,1), (B, n_aux, Tf) fp32 (SURVEY.md 8d synthetic inputs)."""
    rng = np.random.Generator(np.random.PCG64([seed, 77]))
    return rng.standard_normal((batch, cfg.n_aux, n_frames)).astype(np.float32)
