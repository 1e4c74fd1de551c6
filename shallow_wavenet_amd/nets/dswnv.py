"""Drop-in for the reference module `src/nets/dswnv.py` (discrete / softmax mu-law shallow
WaveNet) whose hot paths run as gfx950 HIP kernels.

Import surface `decode_mu_law, encode_mu_law, DSWNV, OneHot, initialize`
(decode_dswnv_softmax.py:28, train_dswnv_softmax.py:30-31); constructor kwargs dswnv.py:191-193;
identical state_dict keys / shapes / default-init order; methods
  * `forward(audio, aux, do=False, last=False)` -> (B, T, Q) logits          dswnv.py:250-276
  * `batch_fast_generate(audio, aux, n_samples_list, intervals=4410)`        dswnv.py:296-399
The one-hot input (dswnv.py:68-93) is never materialised on the device: a one-hot through a conv
is a column gather of its weight, done inside the kernels.
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
if _PKG_PARENT not in sys.path:
    sys.path.insert(0, _PKG_PARENT)

from shallow_wavenet_amd.config import NetConfig                      # noqa: E402
from shallow_wavenet_amd import noise as _noise                       # noqa: E402
from shallow_wavenet_amd.nets._engine import (                        # noqa: E402,F401
    CausalConv1d, EngineMixin, TwoSidedDilConv1d, UpSampling, dropout_device, initialize, log_decode_speed,
    resolve_noise_source)


def encode_mu_law(x, mu=256):
    """float audio in [-1, 1] -> integer classes 0..mu-1 (numpy; dswnv.py:19-31)."""
    m = mu - 1
    fx = np.sign(x) * np.log(1 + m * np.abs(x)) / np.log(1 + m)
    return np.floor((fx + 1) / 2 * m + 0.5).astype(np.int64)


def decode_mu_law(y, mu=256):
    """integer classes -> float audio; keeps the reference's (y - 0.5) offset, so
    decode(0) = -1.0221 and decode(255) = 0.9784 (dswnv.py:34-47)."""
    m = mu - 1
    fx = (y - 0.5) / m * 2 - 1
    return np.sign(fx) / m * ((1 + m) ** np.abs(fx) - 1)


def OneHot(x, depth=256):
    """(B, T) int64 -> (B, T, depth) float one-hot of x % depth, on the GPU when one is present
    (a data-format helper for callers; the kernels take the indices directly)."""
    x = (x % depth).unsqueeze(2)
    out = torch.zeros(x.size(0), x.size(1), depth, dtype=torch.float32)
    if torch.cuda.is_available():
        out, x = out.cuda(), x.cuda()
    return out.scatter_(2, x, 1)


class DSWNV(EngineMixin, nn.Module):
    def __init__(self, n_quantize=256, n_aux=54, hid_chn=192, skip_chn=256, aux_kernel_size=3,
                 aux_dilation_size=2, dilation_depth=3, dilation_repeat=3, kernel_size=6,
                 upsampling_factor=110, audio_in_flag=False, wav_conv_flag=False, do_prob=0):
        super().__init__()
        self.n_aux = n_aux
        self.n_quantize = n_quantize
        self.upsampling_factor = upsampling_factor
        self.in_audio_dim = n_quantize
        self.n_hidch = hid_chn
        self.n_skipch = skip_chn
        self.kernel_size = kernel_size
        self.dilation_depth = dilation_depth
        self.dilation_repeat = dilation_repeat
        self.aux_kernel_size = aux_kernel_size
        self.aux_dilation_size = aux_dilation_size
        self.do_prob = do_prob
        self.audio_in_flag = audio_in_flag
        self.wav_conv_flag = wav_conv_flag
        self._cfg = NetConfig(kind="softmax", n_aux=n_aux, hid_chn=hid_chn, skip_chn=skip_chn,
                              aux_kernel_size=aux_kernel_size, aux_dilation_size=aux_dilation_size,
                              dilation_depth=dilation_depth, dilation_repeat=dilation_repeat,
                              kernel_size=kernel_size, upsampling_factor=upsampling_factor,
                              n_quantize=n_quantize, wav_conv_flag=bool(wav_conv_flag),
                              audio_in_flag=bool(audio_in_flag))

        self.scale_in = nn.Conv1d(n_aux, n_aux, 1)
        self.conv_aux = TwoSidedDilConv1d(in_dim=n_aux, kernel_size=aux_kernel_size, layers=aux_dilation_size)
        self.in_aux_dim = n_aux * self.conv_aux.rec_field
        self.upsampling = UpSampling(upsampling_factor)
        if do_prob > 0:
            self.aux_drop = nn.Dropout(p=do_prob)
        self.in_tot_dim = self.in_aux_dim + (self.in_audio_dim if audio_in_flag else 0)
        if wav_conv_flag:
            self.wav_conv = nn.Conv1d(self.in_audio_dim, hid_chn, 1)
            self.causal = CausalConv1d(hid_chn, hid_chn, kernel_size, dil_fact=0)
        else:
            self.causal = CausalConv1d(self.in_audio_dim, hid_chn, kernel_size, dil_fact=0)

        self.padding = []
        self.dil_facts = [i for i in range(dilation_depth)] * dilation_repeat
        logging.info(self.dil_facts)
        self.in_x = nn.ModuleList()
        self.dil_h = nn.ModuleList()
        self.out_skip = nn.ModuleList()
        for i, d in enumerate(self.dil_facts):
            self.in_x.append(nn.Conv1d(self.in_tot_dim, hid_chn * 2, 1))
            self.dil_h.append(CausalConv1d(hid_chn, hid_chn * 2, kernel_size, dil_fact=d))
            self.padding.append(self.dil_h[i].padding)
            self.out_skip.append(nn.Conv1d(hid_chn, skip_chn, 1))
        logging.info(self.padding)
        self.receptive_field = sum(self.padding) + kernel_size - 1
        logging.info(self.receptive_field)
        if do_prob > 0:
            self.dcrnn_drop = nn.Dropout(p=do_prob)
        self.out_1 = nn.Conv1d(skip_chn, n_quantize, 1)
        self.out_2 = nn.Conv1d(n_quantize, n_quantize, 1)
        assert self.receptive_field == self._cfg.receptive_field

    @staticmethod
    def _indices(audio, depth):
        """accept the reference's one-hot (B, Q, T) float input or plain (B, T) integer classes."""
        if audio.dim() == 3:
            if audio.shape[1] != depth:
                raise RuntimeError(f"one-hot audio must be (B, {depth}, T)")
            return audio.argmax(dim=1)
        return audio % depth

    def forward(self, audio, aux, do=False, last=False):
        """audio: one-hot (B, Q, Tf*U-1) as the training script builds it (or (B, Tf*U-1) classes),
        aux (B, n_aux, Tf) -> logits (B, Tf*U-1, Q)."""
        net = self._engine()
        idx = self._indices(audio, self.n_quantize)
        drop = None
        if do and self.do_prob > 0 and self.training:          # nn.Dropout acts in training mode only
            drop = _noise.dropout_masks(self._cfg, aux.shape[0], aux.shape[2], self.do_prob, device=dropout_device(self))
        if drop is not None or (torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list())):
            from shallow_wavenet_amd.nets._autograd import StackFunction
            self._pending_drop = drop
            raw = StackFunction.apply(self, aux, idx, *self._param_list())
        else:
            raw, _ = net.forward(aux, idx)
        return raw.transpose(1, 2)

    def batch_fast_generate(self, audio, aux, n_samples_list, intervals=4410):
        """audio (B, 1) seed class (encode_mu_law(0) = Q/2 in the decode driver, decode_dswnv_softmax.py:94-99; any
        class is accepted, dswnv.py:305-336), aux (B, n_aux, Tf) -> list of B int64 class arrays trimmed to n_samples.

        Noise (`self.noise_source`): "device" (default for this model) - the Exp(1) deviates of the categorical sampler
        (dswnv.py:361-369) are drawn inside the kernels, keyed by one value taken from the torch CPU generator
        (reproducible under `torch.manual_seed`), like the reference sampling on its model's device; "host" draws them
        with the torch CPU generator in the reference's order - bit-exact indices against the reference's CPU decode,
        at the price of B x n_steps x Q floats drawn, stored and uploaded by the host."""
        with torch.no_grad():
            net = self._engine()
            B = aux.shape[0]
            n_steps = max(n_samples_list)
            start = time.time()
            seed = audio.reshape(B, -1)[:, -1] % self.n_quantize
            seed = seed if torch.count_nonzero(seed - self.n_quantize // 2).item() != 0 else None
            if resolve_noise_source(self, "device") == "host":
                noise = _noise.softmax_exponential(self._cfg, n_steps, B)     # Exp(1) draws of multinomial(n=1)
                out, _ = net.decode(aux, n_steps, noise, seed=seed)
            else:
                # one fresh key per call unless the caller pinned one for the run (decode_driver: so that an utterance's
                # stream depends only on (key, its global index), not on batching or on the number of GPUs)
                key = getattr(self, "noise_rng_seed", None)
                out, _ = net.decode(aux, n_steps, None, seed=seed, rng_seed=_noise.draw_rng_seed() if key is None else int(key),
                                    rng_utt0=int(getattr(self, "noise_utterance_offset", 0)),
                                    utt_ids=getattr(self, "noise_utterance_ids", None))
            samples = out.cpu().numpy().astype(np.int64)
            log_decode_speed(1, n_steps, len(n_samples_list), time.time() - start)
        return [samples[b, :n] for b, n in zip(range(B), n_samples_list)]
