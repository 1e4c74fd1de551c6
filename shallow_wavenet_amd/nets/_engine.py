"""Shared machinery of the two drop-in modules: parameter holders with the reference's
state_dict names, and the cache that keeps a packed copy of the parameters in HBM.

The holders carry parameters only; all arithmetic happens in the HIP kernels reached
through `shallow_wavenet_amd.runtime.HipNet`.  There is no CPU fallback.
"""
from __future__ import annotations

import logging
import time
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from ..config import NetConfig
from ..runtime import HipNet, pack_parameters_device


def initialize(m):
    """Xavier-uniform Conv1d weights with zero bias, unit ConvTranspose weights with zero bias
    (same effect as the reference helper, cswnv_shift1.py:20-34 / dswnv.py:50-64)."""
    if isinstance(m, nn.Conv1d):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0.0)
    if isinstance(m, (nn.ConvTranspose2d, nn.ConvTranspose1d)):
        nn.init.constant_(m.weight, 1.0)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0.0)


class _Holder(nn.Module):
    def forward(self, *a, **k):       # parameters only
        raise RuntimeError("parameter holder: the computation runs in the fused HIP kernels")


class UpSampling(_Holder):
    """holds `conv` = ConvTranspose2d(1,1,(1,U),stride (1,U)) -> keys upsampling.conv.{weight,bias}"""

    def __init__(self, upsampling_factor, bias=True):
        super().__init__()
        self.upsampling_factor = upsampling_factor
        self.bias = bias
        self.conv = nn.ConvTranspose2d(1, 1, kernel_size=(1, upsampling_factor),
                                       stride=(1, upsampling_factor), bias=bias)


class CausalConv1d(_Holder):
    """holds `conv` = Conv1d(Cin, Cout, K, dilation=K**dil_fact); padding = K**(d+1) - K**d"""

    def __init__(self, in_channels, out_channels, kernel_size, dil_fact=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.dil_fact = dil_fact
        self.dilation = kernel_size ** dil_fact
        self.padding = kernel_size ** (dil_fact + 1) - self.dilation
        self.bias = bias
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, padding=self.padding,
                              dilation=self.dilation, bias=bias)


class TwoSidedDilConv1d(_Holder):
    """holds `conv` = ModuleList of the channel-expanding k=3 conditioning convs"""

    def __init__(self, in_dim=39, kernel_size=3, layers=2):
        super().__init__()
        self.in_dim, self.kernel_size, self.layers = in_dim, kernel_size, layers
        self.rec_field = kernel_size ** layers
        self.conv = nn.ModuleList()
        for i in range(layers):
            self.conv.append(nn.Conv1d(in_dim * kernel_size ** i, in_dim * kernel_size ** (i + 1), kernel_size,
                                       stride=1, dilation=kernel_size ** i,
                                       padding=(kernel_size ** (i + 1) - kernel_size ** i) // 2))


class EngineMixin:
    """packed-parameter cache: rebuilt whenever a parameter tensor is replaced or modified in
    place (load_state_dict, optimizer step, .to()/.cuda()); SURVEY.md 8b 'Checkpoint / ownership'."""

    _cfg: NetConfig
    device_unfold = True      # packed gradients -> p.grad by one launch (False: the torch-op version, nets/_autograd.py)

    def _param_slots(self):
        """[(module._parameters, name)] in state_dict order, built once: walking the module tree on every call costs more
        host time than a BL6 layer kernel runs.  Reading the Parameter through its slot sees a replaced object; a module
        ADDED after the first call is not seen (the reference's models are fixed after construction)."""
        slots = self.__dict__.get("_slots")
        if slots is None:
            slots = [(m._parameters, k) for m in self.modules() for k, v in m._parameters.items() if v is not None]
            self.__dict__["_slots"] = slots
        return slots

    def _param_list(self):
        return [d[k] for d, k in self._param_slots()]

    def _engine_key(self):
        ps = self._param_list()
        return (ps[0].device, tuple((p.data_ptr(), p._version) for p in ps))

    def _engine(self) -> HipNet:
        ps = self._param_list()
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError(
                f"{type(self).__name__}: parameters are on {dev}; the MI355X build has no CPU path - "
                "call model.cuda() on a machine with a HIP device")
        key = (dev, tuple((p.data_ptr(), p._version) for p in ps))
        cache = self.__dict__.get("_engine_cache")
        if cache is None or cache[0] != key:
            tensors = ps                       # Module.parameters() order is state_dict order (no buffers in these models)
            if cache is not None and cache[1].device == dev:
                # parameters changed (optimizer step, load_state_dict): re-lay them out on the device, in place
                net = cache[1]
                net.repack(tensors)
            else:
                net = HipNet(self._cfg, pack_parameters_device(self._cfg, tensors), dev)
            self.__dict__["_engine_cache"] = (key, net)
            return net
        return cache[1]

    def set_packed_engine(self, net: HipNet) -> None:
        """install an engine whose packed buffer arrived by RCCL broadcast (dist.py)."""
        self.__dict__["_engine_cache"] = (self._engine_key(), net)


NOISE_SOURCES = ("host", "device")


def dropout_device(module):
    """where the nn.Dropout masks of a training-mode forward are drawn: `module.dropout_source` = "device" (default: the
    module's own device, like nn.Dropout of the reference on a GPU) | "host" (torch CPU generator in the reference's draw
    order: reproduces a CPU run of the reference mask for mask under torch.manual_seed - the g5_drop_* fixtures)."""
    src = getattr(module, "dropout_source", None) or "device"
    if src not in NOISE_SOURCES:
        raise ValueError(f"dropout_source must be one of {NOISE_SOURCES} or None, not {src!r}")
    return None if src == "host" else module._param_list()[0].device


def resolve_noise_source(module, default: str) -> str:
    """`module.noise_source`: "host" | "device" | None (= the model's default)."""
    src = getattr(module, "noise_source", None) or default
    if src not in NOISE_SOURCES:
        raise ValueError(f"noise_source must be one of {NOISE_SOURCES} or None, not {src!r}")
    return src


def log_decode_speed(seg: int, n_steps: int, n_utts: int, seconds: float) -> None:
    """the two summary lines of the reference loop (cswnv_shift1.py:417-422 / dswnv.py:386-391);
    per-step times are not observable from inside one persistent launch, so the mean is used."""
    n = max(n_steps * seg, 1)
    per = seconds / n
    logging.info("average time / sample = %.6f sec (%ld samples) [%.3f kHz/s]" % (per, n, 1.0 / (1000 * per)))
    logging.info("average throughput / sample = %.6f sec (%ld samples * %ld) [%.3f kHz/s]" % (
        seconds / (n * n_utts), n, n_utts, n * n_utts / (1000 * seconds)))
