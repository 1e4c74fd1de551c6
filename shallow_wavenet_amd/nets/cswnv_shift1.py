"""Drop-in for the reference module `src/nets/cswnv_shift1.py` (continuous / Laplace shallow
WaveNet) whose hot paths run as gfx950 HIP kernels.

Same import surface as the reference (`CSWNV, LaplaceLoss, LSDloss, initialize` plus the
holder classes), same constructor kwargs (cswnv_shift1.py:131-133), same sub-module names and
therefore identical `state_dict()` keys/shapes and default-initialisation RNG order, same
method signatures and return values:

  * `forward(aux, audio, do=False, clip=False)`            cswnv_shift1.py:191-267
  * `batch_fast_generate(audio, aux, n_samples_list, intervals=4410, Laplace=True)`   :287-430

What differs on purpose: the math never runs in torch.  `forward` and
`batch_fast_generate` pack the parameters once (cached until a parameter changes), run the
frame-rate front end, and call the decode / teacher-forced kernels through the C ABI
(`include/swn_hip.h`).  Noise is drawn by the host torch CPU generator in the reference's
draw order, so `torch.manual_seed(s)` reproduces the reference's CPU results.
"""
from __future__ import annotations

import logging
import os
import sys
import time

import numpy as np
import torch
from torch import nn

_PKG_PARENT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _PKG_PARENT not in sys.path:            # importable as top-level `cswnv_shift1` (PYTHONPATH=.../nets, path.sh:9)
    sys.path.insert(0, _PKG_PARENT)

from shallow_wavenet_amd.config import NetConfig                      # noqa: E402
from shallow_wavenet_amd import noise as _noise                       # noqa: E402
from shallow_wavenet_amd.nets._engine import (                        # noqa: E402,F401
    CausalConv1d, EngineMixin, TwoSidedDilConv1d, UpSampling, dropout_device, initialize, log_decode_speed,
    resolve_noise_source)


class CSWNV(EngineMixin, nn.Module):
    def __init__(self, n_aux=54, hid_chn=192, skip_chn=256, aux_kernel_size=3, aux_dilation_size=2,
                 dilation_depth=3, dilation_repeat=2, kernel_size=7, upsampling_factor=110, seg=5, lpc=4,
                 do_prob=0, aux_conv2d_flag=False, wav_conv_flag=False):
        super().__init__()
        self.n_aux = n_aux
        self.n_hidch = hid_chn
        self.n_skipch = skip_chn
        self.aux_kernel_size = aux_kernel_size
        self.aux_dilation_size = aux_dilation_size
        self.dilation_depth = dilation_depth
        self.dilation_repeat = dilation_repeat
        self.kernel_size = kernel_size
        self.upsampling_factor = upsampling_factor
        self.seg = seg
        self.lpc = lpc
        self.lpc_offset = seg - lpc
        self.do_prob = do_prob
        self.aux_conv2d_flag = aux_conv2d_flag
        self.wav_conv_flag = wav_conv_flag
        self._cfg = NetConfig(kind="laplace", n_aux=n_aux, hid_chn=hid_chn, skip_chn=skip_chn,
                              aux_kernel_size=aux_kernel_size, aux_dilation_size=aux_dilation_size,
                              dilation_depth=dilation_depth, dilation_repeat=dilation_repeat,
                              kernel_size=kernel_size, upsampling_factor=upsampling_factor, seg=seg, lpc=lpc,
                              wav_conv_flag=bool(wav_conv_flag), aux_conv2d_flag=bool(aux_conv2d_flag))

        # parameter holders, created in the reference's construction order so that a seeded
        # default construction draws identical initial values
        self.scale_in = nn.Conv1d(n_aux, n_aux, 1)
        self.conv_aux = TwoSidedDilConv1d(in_dim=n_aux, kernel_size=aux_kernel_size, layers=aux_dilation_size)
        self.upsampling = UpSampling(upsampling_factor)
        if do_prob > 0:
            self.aux_drop = nn.Dropout(p=do_prob)
        self.in_aux_dim = n_aux * self.conv_aux.rec_field
        if aux_conv2d_flag and seg > 1:
            self.aux_conv2d = nn.Conv2d(self.in_aux_dim, self.in_aux_dim, (seg, 1))
        elif seg > 1:
            self.in_aux_dim *= seg
        if wav_conv_flag:
            self.wav_conv = nn.Conv1d(1, hid_chn, 1)
            self.causal = CausalConv1d(hid_chn, hid_chn, kernel_size, dil_fact=0)
        else:
            self.causal = CausalConv1d(1, hid_chn, kernel_size, dil_fact=0)

        self.padding = []
        self.dil_facts = [i for i in range(dilation_depth)] * dilation_repeat
        logging.info(self.dil_facts)
        self.in_x = nn.ModuleList()
        self.dil_h = nn.ModuleList()
        self.out_skip = nn.ModuleList()
        for i, d in enumerate(self.dil_facts):
            self.in_x.append(nn.Conv1d(self.in_aux_dim, hid_chn * 2, 1))
            self.dil_h.append(CausalConv1d(hid_chn, hid_chn * 2, kernel_size, dil_fact=d))
            self.padding.append(self.dil_h[i].padding)
            self.out_skip.append(nn.Conv1d(hid_chn, skip_chn, 1))
        logging.info(self.padding)
        self.receptive_field = sum(self.padding) + kernel_size - 1
        logging.info(self.receptive_field)
        if do_prob > 0:
            self.dcrnn_drop = nn.Dropout(p=do_prob)
        self.out_1 = nn.Conv1d(skip_chn, skip_chn, 1)
        self.out_2 = nn.Conv1d(skip_chn, 2 * seg + lpc, 1)
        assert self.receptive_field == self._cfg.receptive_field

    # ------------------------------------------------------------------ teacher-forced stack
    def forward(self, aux, audio, do=False, clip=False):
        """aux (B, n_aux, Tf), audio (B, 1, Tf*U - seg) -> the reference's tuples:
        lpc>0: (mu, b, log_b, a) | clip: (mu, b_noclip, b, log_b, a); lpc==0 drops `a`;
        seg==1 and lpc==0 returns 2-D (B, T') tensors (cswnv_shift1.py:228-267)."""
        net = self._engine()
        # nn.Dropout only acts in training mode (model.train(), train_cswnv...py:716); masks are drawn on the host
        # in the reference's order and handed to the dropout-mode kernels
        drop = None
        # a <=2-layer stack sends layer 0 through dcrnn_drop whatever `do` says (cswnv_shift1.py:220-223); aux_drop
        # still follows `do`
        two = self.dilation_depth * self.dilation_repeat <= 2
        if self.do_prob > 0 and self.training and (do or two):
            drop = _noise.dropout_masks(self._cfg, aux.shape[0], aux.shape[2], self.do_prob, draw_x=bool(do),
                                        device=dropout_device(self))
        if drop is not None or (torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list())):
            # training: HIP forward + HIP backward behind autograd Functions (nets/_autograd.py)
            from shallow_wavenet_amd.nets._autograd import LaplaceHeadFunction, StackFunction
            self._pending_drop = drop
            raw = StackFunction.apply(self, aux, audio, *self._param_list())
            mu, b, log_b, a, b_clip, log_b_clip, flag = LaplaceHeadFunction.apply(net, raw, clip)
            a = a if self.lpc > 0 else None
            if not clip:
                b_clip = log_b_clip = None
        else:
            # opt-in: `model.bf16_forward = True` evaluates the stack with the bf16 MFMA kernels (fp32 accumulation;
            # outputs differ from the fp32 path by ~5e-4) - evaluation passes that do not need the 1e-5 parity
            raw = net.forward_bf16(aux, audio) if getattr(self, "bf16_forward", False) else net.forward(aux, audio)[0]
            mu, b, log_b, a, b_clip, log_b_clip, flag = net.laplace_head(raw, clip=clip)
        if self.lpc == 0 and self.seg == 1:
            sq = lambda x: None if x is None else x.reshape(x.shape[0], -1)
            mu, b, log_b, b_clip, log_b_clip = sq(mu), sq(b), sq(log_b), sq(b_clip), sq(log_b_clip)
        tail = (a,) if self.lpc > 0 else ()
        if not clip:
            return (mu, b, log_b) + tail
        if int(flag.item()) != 0:                       # torch.min(log_b) < floor, :234
            return (mu, b, b_clip, log_b_clip) + tail
        return (mu, b, b, log_b) + tail

    # ------------------------------------------------------------------ autoregressive decode
    def batch_fast_generate(self, audio, aux, n_samples_list, intervals=4410, Laplace=True):
        """audio (B, seg) seed waveform (zeros in the decode driver, decode_cswnv_laplace-shift1.py:93; any values are
        accepted, cswnv_shift1.py:300-334), aux (B, n_aux, Tf) zero-padded features -> list of B float32 arrays
        trimmed to n_samples.

        Noise (`self.noise_source`): "host" (default for this model) draws the uniform deviates with the torch CPU
        generator in the reference's order, so `torch.manual_seed(s)` reproduces the reference's CPU decode sample for
        sample; "device" lets the kernels draw them (keyed by one 63-bit value taken from the torch CPU generator, so
        runs stay reproducible under `torch.manual_seed`), like the reference drawing on its model's device."""
        if not Laplace:
            raise NotImplementedError("the reference has no non-Laplace branch in this loop either")
        with torch.no_grad():
            net = self._engine()
            B = aux.shape[0]
            max_samples = max(n_samples_list)
            n_steps = int(max_samples / self.seg) if self.seg > 1 else max_samples
            start = time.time()
            seed = audio.reshape(B, -1)[:, -self.seg:] if torch.count_nonzero(audio).item() != 0 else None
            if resolve_noise_source(self, "host") == "host":
                noise = _noise.laplace_uniform(self._cfg, n_steps, B)      # host CPU generator, reference order
                out, _ = net.decode(aux, n_steps, noise, seed=seed)
            else:
                # one fresh key per call unless the caller pinned one for the run (decode_driver: so that an utterance's
                # stream depends only on (key, its global index), not on batching or on the number of GPUs)
                key = getattr(self, "noise_rng_seed", None)
                out, _ = net.decode(aux, n_steps, None, seed=seed, rng_seed=_noise.draw_rng_seed() if key is None else int(key),
                                    rng_utt0=int(getattr(self, "noise_utterance_offset", 0)),
                                    utt_ids=getattr(self, "noise_utterance_ids", None))
            samples = out.cpu().numpy()                                   # DEVICE -> HOST, :426
            log_decode_speed(self.seg, n_steps, len(n_samples_list), time.time() - start)
        samples = samples[:, -max_samples:] if max_samples <= samples.shape[1] else samples
        return [samples[b, :n] for b, n in zip(range(B), n_samples_list)]


class LaplaceLoss(nn.Module):
    """mean Laplace negative log-likelihood  ln2 + log b + |t - mu| / b  with the optional scale
    floor b >= 7.07e-7 (training-side row f2 of SURVEY.md section 8; cswnv_shift1.py:433-453)."""

    def __init__(self):
        super().__init__()
        self.c = 0.69314718055994530941723212145818

    def forward(self, mu, b, target, log_b=None, clip=False, log=True):
        floor_b, floor_lb = 7.0710678118654752440084436210504e-7, -14.162084148244246758816564788835
        if log_b is None:
            if clip and torch.min(b) < floor_b:
                b = torch.clamp(b, min=floor_b)
            log_b = torch.log(b)
        elif clip and torch.min(log_b) < floor_lb:
            log_b = torch.clamp(log_b, min=floor_lb)
            b = torch.exp(log_b)
        if log:
            var = 2 * (b ** 2)
            logging.info("%lf %E %lf %E %E %E %E" % (torch.min(mu), torch.mean(mu), torch.max(mu), torch.var(mu),
                                                    torch.min(var), torch.mean(var), torch.max(var)))
        return torch.mean(self.c + log_b + torch.abs(target - mu) / b)


class LSDloss(nn.Module):
    """log-spectral-distance style losses on (.., bins) magnitudes (cswnv_shift1.py:456-476)."""

    def forward(self, x, y, LSD=True, L2=True):
        if LSD:
            pow_x, pow_y = torch.sum(x ** 2, -1), torch.sum(y ** 2, -1)
            if L2:
                return torch.mean(torch.sqrt(torch.mean((10 * (torch.log10(pow_x) - torch.log10(pow_y))) ** 2, 0)))
            return torch.mean((pow_x - pow_y) ** 2)
        return torch.mean((x - y) ** 2 if L2 else torch.abs(x - y))
