"""CPU oracle for the shallow-WaveNet hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU (PyTorch-CPU, fp32) restatement of the reference algorithm.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it; the product package `shallow_wavenet_amd` never does.  It is the *checker*, never the
thing measured or shipped.

Parity pinning: the reference has no tests or golden vectors of its own (SURVEY.md
section 4), so this restatement is pinned against fixtures produced by importing the
reference's own `src/nets` modules in the build container (`oracle/make_golden.py`,
fixtures under `tests/golden/`); `tests/test_oracle_golden.py` checks every one of them.

Functions are functional (no nn.Module): `P` is a dict of fp32 CPU tensors keyed with the
reference's state_dict names, `cfg` a `shallow_wavenet_amd.config.NetConfig`.
Noise is always an explicit input, laid out (n_steps, B, seg) for the Laplace model and
(n_steps, B, Q) for the softmax model; `laplace_noise` / `softmax_noise` draw it from the
host torch CPU generator in the reference's draw order.

Reference lines followed (relative to /root/reference/src/nets):
  frontend            cswnv_shift1.py:95-127,152-155,193,297   dswnv.py:155-187,252,302
  upsample            cswnv_shift1.py:37-65
  causal_conv         cswnv_shift1.py:68-92
  gated_layer         cswnv_shift1.py:275-285                   dswnv.py:284-294
  laplace_forward     cswnv_shift1.py:191-267
  laplace_generate    cswnv_shift1.py:287-430
  softmax_forward     dswnv.py:250-276
  softmax_generate    dswnv.py:296-399
  mu-law / one-hot    dswnv.py:19-47,68-93
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LOG_B_FLOOR = -14.162084148244246758816564788835   # cswnv_shift1.py:234


# --------------------------------------------------------------------------- helpers
def as_params(sd: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in sd.items()}


def encode_mu_law(x, mu: int = 256):
    """dswnv.py:19-31 (numpy float64 -> int64)."""
    m = mu - 1
    fx = np.sign(x) * np.log(1 + m * np.abs(x)) / np.log(1 + m)
    return np.floor((fx + 1) / 2 * m + 0.5).astype(np.int64)


def decode_mu_law(y, mu: int = 256):
    """dswnv.py:34-47; note the (y - 0.5) offset: decode(0) = -1.0221."""
    m = mu - 1
    fx = (y - 0.5) / m * 2 - 1
    return np.sign(fx) / m * ((1 + m) ** np.abs(fx) - 1)


def one_hot(idx: torch.Tensor, depth: int) -> torch.Tensor:
    """dswnv.py:68-93: (B,T) int64 -> (B,T,depth) fp32, applying idx % depth."""
    idx = idx % depth
    out = torch.zeros(idx.shape[0], idx.shape[1], depth, dtype=torch.float32)
    return out.scatter_(2, idx.unsqueeze(2), 1.0)


def laplace_noise(cfg, n_steps: int, batch: int, generator: Optional[torch.Generator] = None
                  ) -> np.ndarray:
    """uniform(-0.4999, 0.5) in the reference's draw order -> (n_steps, B, seg).

    lpc > 0 : one (B,1,1) draw per generated sample, order (step, j, b)  cswnv_shift1.py:373,380
    lpc == 0: one (B,1,seg) draw per step, order (step, b, j)            cswnv_shift1.py:387
    CPU uniform_ is a serial kernel, so one big draw equals the sequence of small draws.
    """
    if cfg.lpc > 0:
        e = torch.empty(n_steps, cfg.seg, batch).uniform_(-0.4999, 0.5, generator=generator)
        return e.permute(0, 2, 1).contiguous().numpy()
    e = torch.empty(n_steps, batch, cfg.seg).uniform_(-0.4999, 0.5, generator=generator)
    return e.numpy()


def softmax_noise(cfg, n_steps: int, batch: int, generator: Optional[torch.Generator] = None
                  ) -> np.ndarray:
    """Exp(1) draws of torch.multinomial's n=1 path, one (B,Q) draw per step (dswnv.py:363-365).
    Drawn per step with the same shape the sampler uses so the generator stream matches."""
    out = np.empty((n_steps, batch, cfg.n_quantize), dtype=np.float32)
    for i in range(n_steps):
        out[i] = torch.empty(batch, cfg.n_quantize).exponential_(1, generator=generator).numpy()
    return out


# --------------------------------------------------------------------------- building blocks
def frontend(cfg, P, aux: torch.Tensor) -> torch.Tensor:
    """scale_in (1x1) then the two-sided dilated k=3 stack: (B,n_aux,Tf) -> (B,A0,Tf)."""
    c = F.conv1d(aux, P["scale_in.weight"], P["scale_in.bias"])
    k = cfg.aux_kernel_size
    for i in range(cfg.aux_dilation_size):
        c = F.conv1d(c, P[f"conv_aux.conv.{i}.weight"], P[f"conv_aux.conv.{i}.bias"],
                     dilation=k ** i, padding=(k ** (i + 1) - k ** i) // 2)
    return c


def upsample(cfg, P, c: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d(1,1,(1,U),stride (1,U)) == rank-1 expansion y[c,f*U+j] = x[c,f]*w[j]+b."""
    return F.conv_transpose2d(c.unsqueeze(1), P["upsampling.conv.weight"],
                              P["upsampling.conv.bias"], stride=(1, cfg.U)).squeeze(1)


def causal_conv(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, dil: int) -> torch.Tensor:
    """left-zero-padded dilated conv, output cropped to the input length."""
    k = w.shape[-1]
    return F.conv1d(x, w, b, padding=(k - 1) * dil, dilation=dil)[:, :, : x.shape[2]]


def gated_layer(cfg, P, l: int, x: torch.Tensor, h: torch.Tensor, n_last: Optional[int] = None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """one DCRNN layer; n_last=None is the teacher-forced form, n_last=n the incremental one
    (only the last n columns of the dilated conv and of h are used)."""
    H = cfg.H
    a = causal_conv(h, P[f"dil_h.{l}.conv.weight"], P[f"dil_h.{l}.conv.bias"], cfg.dilations[l])
    if n_last is not None:
        a = a[:, :, -n_last:]
        h = h[:, :, -n_last:]
    g = F.conv1d(x, P[f"in_x.{l}.weight"], P[f"in_x.{l}.bias"]) * a
    z = torch.sigmoid(g[:, :H])
    hn = (1 - z) * torch.tanh(g[:, H:]) + z * h
    return F.conv1d(hn, P[f"out_skip.{l}.weight"], P[f"out_skip.{l}.bias"]), hn


def _stack_seg(cfg, P, x: torch.Tensor) -> torch.Tensor:
    """seg > 1: stack seg consecutive conditioning columns into channels (index c*seg+s),
    or run the optional (seg,1) Conv2d.  cswnv_shift1.py:196-201 / :306-311."""
    if cfg.seg <= 1:
        return x
    u = x.unfold(2, cfg.seg, 1)                       # B, C, T', seg
    if cfg.aux_conv2d_flag:
        return F.conv2d(u.permute(0, 1, 3, 2), P["aux_conv2d.weight"], P["aux_conv2d.bias"]).squeeze(2)
    return u.permute(0, 2, 1, 3).reshape(u.shape[0], u.shape[2], -1).permute(0, 2, 1)


def head(cfg, P, skip_sum: torch.Tensor) -> torch.Tensor:
    y = F.conv1d(F.relu(skip_sum), P["out_1.weight"], P["out_1.bias"])
    return F.conv1d(F.relu(y), P["out_2.weight"], P["out_2.bias"])


def _lift(cfg, P, audio: torch.Tensor) -> torch.Tensor:
    if cfg.wav_conv_flag:
        return F.conv1d(audio, P["wav_conv.weight"], P["wav_conv.bias"])
    return audio


# --------------------------------------------------------------------------- Laplace model
def laplace_stack(cfg, P, aux: torch.Tensor, audio: torch.Tensor, drop=None):
    """raw out_2 output (B, n_out, T-2seg+1) of the teacher-forced stack plus the per-layer
    hidden states (used by kernel-level tests).  drop = (drop_x, [mask or None per layer]) restates the
    training-mode dropout of cswnv_shift1.py:194-195,211-217,269-273 with explicit multiplicative masks:
    `aux_drop` on the upsampled conditioning, `dcrnn_drop` on the hidden state a dropped layer passes on
    (its skip output uses the undropped state)."""
    seg = cfg.seg
    x = upsample(cfg, P, frontend(cfg, P, aux))[:, :, seg:]
    if drop is not None:
        x = x * drop[0]
    x = _stack_seg(cfg, P, x)
    h = F.softsign(causal_conv(_lift(cfg, P, audio), P["causal.conv.weight"],
                               P["causal.conv.bias"], 1)[:, :, seg - 1:])
    hs = [h]
    tot = None
    for l in range(cfg.L):
        sk, h = gated_layer(cfg, P, l, x, h)
        hs.append(h)
        if drop is not None and drop[1][l] is not None:
            h = h * drop[1][l]
        tot = sk if tot is None else tot + sk
    return head(cfg, P, tot), hs


def laplace_forward(cfg, P, aux: torch.Tensor, audio: torch.Tensor, clip: bool = False, drop=None):
    """CSWNV.forward(aux, audio, do, clip) - same return tuples as the reference; drop = explicit training-mode
    masks as in laplace_stack (None = do=False / eval)."""
    seg = cfg.seg
    out, _ = laplace_stack(cfg, P, aux, audio, drop=drop)
    out = out.transpose(1, 2)
    mu = out[:, :, :seg]
    log_b = F.logsigmoid(out[:, :, seg:2 * seg])
    if cfg.lpc == 0 and seg == 1:
        mu = mu.reshape(out.shape[0], -1)
        log_b = log_b.reshape(out.shape[0], -1)
    tail = (out[:, :, 2 * seg:],) if cfg.lpc > 0 else ()
    if not clip:
        return (mu, torch.exp(log_b), log_b) + tail
    b_noclip = torch.exp(log_b)
    if torch.min(log_b) < LOG_B_FLOOR:
        log_b = torch.clamp(log_b, min=LOG_B_FLOOR)
        return (mu, b_noclip, torch.exp(log_b), log_b) + tail
    return (mu, b_noclip, b_noclip, log_b) + tail


def laplace_transform(eps: torch.Tensor) -> torch.Tensor:
    """uniform eps -> unit Laplace deviate (sign folded in): -sign(e)*log1p(-2|e|)."""
    return -eps.sign() * torch.log1p(-2 * eps.abs())


def laplace_generate(cfg, P, aux: torch.Tensor, n_samples_list: Sequence[int], noise,
                     return_heads: bool = False, seed=None):
    """CSWNV.batch_fast_generate; `seed` = its `audio` argument (B, seg), zeros when None (the decode driver's
    seed, decode_cswnv_laplace-shift1.py:93): left-padded with rf zeros, it is the newest part of the first causal
    window and of the LP buffer (cswnv_shift1.py:300-319).

    Keeps the reference's per-step op structure (window convolutions per layer, history
    buffers concatenated and slid) so that timing it is a fair CPU baseline; the growing
    torch.cat of the reference (:394-402) is replaced by writes into preallocated history,
    which only makes this baseline faster than the original.
    """
    seg, lpc, K, L = cfg.seg, cfg.lpc, cfg.K, cfg.L
    B = aux.shape[0]
    rf = cfg.receptive_field
    n_steps = int(max(n_samples_list) / seg) if seg > 1 else max(n_samples_list)
    noise = torch.as_tensor(np.asarray(noise), dtype=torch.float32)
    assert noise.shape[0] >= n_steps and noise.shape[1] == B and noise.shape[2] == seg

    with torch.no_grad():
        x = upsample(cfg, P, frontend(cfg, P, aux))
        x = _stack_seg(cfg, P, F.pad(x, (rf, 0), "replicate"))
        audio = torch.zeros(B, 1, rf + seg)
        if seed is not None:
            audio[:, 0, rf:] = torch.as_tensor(seed, dtype=torch.float32).reshape(B, seg)
        n0 = audio.shape[-1] - (seg - 1)                 # rf + 1 prologue positions
        lp_buf = audio[:, :, seg - 1:][:, :, -lpc:].clone() if lpc > 0 else None      # :317-319

        # lifted sample history, preallocated: prologue part then one slot per new sample
        lifted0 = _lift(cfg, P, audio)
        hist = torch.empty(B, lifted0.shape[1], lifted0.shape[2] + n_steps * seg)
        hist[:, :, : lifted0.shape[2]] = lifted0
        n_hist = lifted0.shape[2]

        # prologue: teacher-forced stack over the rf+1 seed positions, keep layer histories
        h = F.softsign(causal_conv(lifted0, P["causal.conv.weight"], P["causal.conv.bias"], 1)[:, :, seg - 1:])
        x0 = x[:, :, :n0]
        buf_len = [cfg.paddings[l + 1] if l < L - 1 else K - 1 for l in range(L)]
        bufs = []
        for l in range(L):
            _, h = gated_layer(cfg, P, l, x0, h)
            bufs.append(h[:, :, -buf_len[l] - seg: -seg])

        win_out = K + seg - 1
        win_in = win_out + K - 1
        out_samples = torch.empty(B, n_steps * seg)
        heads = torch.empty(n_steps, B, cfg.n_out) if return_heads else None
        for i in range(n_steps):
            pos = n_hist - (seg - 1)                     # == samples.size(-1) of the reference
            xi = x[:, :, pos - seg: pos]
            lo = max(n_hist - win_in, seg - 1)            # `samples` starts at audio[seg-1:]
            h = F.softsign(causal_conv(hist[:, :, lo:n_hist], P["causal.conv.weight"],
                                       P["causal.conv.bias"], 1)[:, :, -win_out:])
            tot = None
            for l in range(L):
                sk, h = gated_layer(cfg, P, l, xi, h, n_last=seg)
                h = torch.cat((bufs[l], h), 2)
                bufs[l] = h[:, :, -buf_len[l]:]
                tot = sk if tot is None else tot + sk
            o = head(cfg, P, tot).transpose(1, 2)[:, -1:, :]          # B,1,n_out
            if return_heads:
                heads[i] = o[:, 0, :]
            mu = o[:, :, :seg]
            b = torch.exp(F.logsigmoid(o[:, :, seg:2 * seg]))
            if lpc > 0:
                a = o[:, :, 2 * seg:].flip(-1)
                new = []
                for j in range(seg):
                    e = noise[i, :, j].reshape(B, 1, 1)
                    s = torch.clamp((a * lp_buf).sum(-1, keepdim=True) + mu[:, :, j:j + 1]
                                    - b[:, :, j:j + 1] * e.sign() * torch.log1p(-2 * e.abs()),
                                    min=-1, max=1)
                    lp_buf = torch.cat((lp_buf[:, :, 1:], s), 2)
                    new.append(s)
                new = torch.cat(new, 2)
            else:
                e = noise[i].reshape(B, 1, seg)
                new = torch.clamp(mu - b * e.sign() * torch.log1p(-2 * e.abs()), min=-1, max=1)
            new = new.reshape(B, 1, -1)
            out_samples[:, i * seg:(i + 1) * seg] = new[:, 0]
            hist[:, :, n_hist:n_hist + seg] = _lift(cfg, P, new)
            n_hist += seg

    arr = out_samples.numpy()
    res = [arr[b, :n] for b, n in zip(range(B), n_samples_list)]
    if return_heads:
        return res, heads.numpy()
    return res


# --------------------------------------------------------------------------- softmax model
def softmax_stack(cfg, P, audio_idx: torch.Tensor, aux: torch.Tensor, drop=None):
    """teacher-forced logits (B, Q, T) and hidden states; audio_idx int64 (B, T); drop as in laplace_stack
    (dswnv.py:253-254,264-270)."""
    oh = one_hot(audio_idx, cfg.n_quantize).transpose(1, 2)          # B,Q,T
    x = upsample(cfg, P, frontend(cfg, P, aux))[:, :, 1:]
    if drop is not None:
        x = x * drop[0]
    if cfg.audio_in_flag:
        x = torch.cat((x, oh), 1)
    h = F.softsign(causal_conv(_lift(cfg, P, oh), P["causal.conv.weight"], P["causal.conv.bias"], 1))
    hs = [h]
    tot = None
    for l in range(cfg.L):
        sk, h = gated_layer(cfg, P, l, x, h)
        hs.append(h)
        if drop is not None and drop[1][l] is not None:
            h = h * drop[1][l]
        tot = sk if tot is None else tot + sk
    return head(cfg, P, tot), hs


def softmax_forward(cfg, P, audio_onehot: torch.Tensor, aux: torch.Tensor) -> torch.Tensor:
    """DSWNV.forward(audio, aux): audio is the one-hot (B,Q,T) tensor the training script
    passes (train_dswnv_softmax.py builds it with OneHot); returns (B,T,Q) logits."""
    x = upsample(cfg, P, frontend(cfg, P, aux))[:, :, 1:]
    if cfg.audio_in_flag:
        x = torch.cat((x, audio_onehot), 1)
    h = F.softsign(causal_conv(_lift(cfg, P, audio_onehot), P["causal.conv.weight"],
                               P["causal.conv.bias"], 1))
    tot = None
    for l in range(cfg.L):
        sk, h = gated_layer(cfg, P, l, x, h)
        tot = sk if tot is None else tot + sk
    return head(cfg, P, tot).transpose(1, 2)


def categorical_from_noise(logits: torch.Tensor, q: torch.Tensor):
    """softmax -> OneHotCategorical(probs).sample() -> argmax, with the Exp(1) draws q given:
    Categorical renormalises probs, multinomial(n=1) returns argmax(probs / q)."""
    p = F.softmax(logits, dim=-1)
    p = p / p.sum(-1, keepdim=True)
    r = p / q
    idx = torch.argmax(r, dim=-1)
    top2 = torch.topk(r, 2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1]) / top2[..., 0]
    return idx, margin


def softmax_generate(cfg, P, aux: torch.Tensor, n_samples_list: Sequence[int], noise,
                     seed_index: Optional[int] = None, return_heads: bool = False, seed=None):
    """DSWNV.batch_fast_generate; `seed` = its `audio` argument (B,) of classes (or one class `seed_index` for all),
    encode_mu_law(0) = Q//2 when None; the left padding is always class Q//2 (dswnv.py:308)."""
    K, L, Q = cfg.K, cfg.L, cfg.n_quantize
    B = aux.shape[0]
    rf = cfg.receptive_field
    n_steps = max(n_samples_list)
    noise = torch.as_tensor(np.asarray(noise), dtype=torch.float32)
    seed_index = Q // 2 if seed_index is None else seed_index
    with torch.no_grad():
        x = F.pad(upsample(cfg, P, frontend(cfg, P, aux)), (rf, 0), "replicate")
        audio = torch.full((B, rf + 1), Q // 2, dtype=torch.int64)
        audio[:, -1] = seed_index if seed is None else torch.as_tensor(seed, dtype=torch.int64).reshape(B)
        oh = one_hot(audio, Q).transpose(1, 2)                          # B,Q,rf+1
        x0 = x[:, :, : oh.shape[2]]
        if cfg.audio_in_flag:
            x0 = torch.cat((x0, oh), 1)
        lifted0 = _lift(cfg, P, oh)
        hist = torch.empty(B, lifted0.shape[1], lifted0.shape[2] + n_steps)
        hist[:, :, : lifted0.shape[2]] = lifted0
        n_hist = lifted0.shape[2]
        last_oh = oh[:, :, -1:]
        h = F.softsign(causal_conv(lifted0, P["causal.conv.weight"], P["causal.conv.bias"], 1))
        buf_len = [cfg.paddings[l + 1] if l < L - 1 else K - 1 for l in range(L)]
        bufs = []
        for l in range(L):
            _, h = gated_layer(cfg, P, l, x0, h)
            bufs.append(h[:, :, -buf_len[l] - 1: -1])
        out_idx = torch.empty(B, n_steps, dtype=torch.int64)
        heads = torch.empty(n_steps, B, Q) if return_heads else None
        margins = torch.empty(n_steps, B)
        win = 2 * K - 1
        for i in range(n_steps):
            xi = x[:, :, n_hist - 1: n_hist]
            if cfg.audio_in_flag:
                xi = torch.cat((xi, last_oh), 1)
            lo = max(n_hist - win, 0)
            h = F.softsign(causal_conv(hist[:, :, lo:n_hist], P["causal.conv.weight"],
                                       P["causal.conv.bias"], 1)[:, :, -K:])
            tot = None
            for l in range(L):
                sk, h = gated_layer(cfg, P, l, xi, h, n_last=1)
                h = torch.cat((bufs[l], h), 2)
                bufs[l] = h[:, :, -buf_len[l]:]
                tot = sk if tot is None else tot + sk
            logits = head(cfg, P, tot).transpose(1, 2)[:, -1]           # B,Q
            if return_heads:
                heads[i] = logits
            idx, margins[i] = categorical_from_noise(logits, noise[i])
            out_idx[:, i] = idx
            last_oh = one_hot(idx.unsqueeze(1), Q).transpose(1, 2)
            hist[:, :, n_hist:n_hist + 1] = _lift(cfg, P, last_oh)
            n_hist += 1
    arr = out_idx.numpy()
    res = [arr[b, :n] for b, n in zip(range(B), n_samples_list)]
    if return_heads:
        return res, heads.numpy(), margins.numpy()
    return res


# --------------------------------------------------------------------------- losses (f2)
def laplace_nll(mu, b, target, log_b=None):
    """LaplaceLoss.forward without clipping/logging: mean(ln2 + log b + |t-mu|/b)
    (cswnv_shift1.py:433-453)."""
    if log_b is None:
        log_b = torch.log(b)
    return torch.mean(0.69314718055994530941723212145818 + log_b + torch.abs(target - mu) / b)


# --------------------------------------------------------------------------- in-kernel noise generator (restated)
def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., SC'11) on numpy uint32 arrays: counter (..., 4), key (2,) -> (..., 4).
    Restates csrc/swn_noise.hpp::swn_philox4x32_10; checked against the Random123 known-answer vectors in
    tests/test_oracle_golden.py."""
    c = np.array(counter, dtype=np.uint64, copy=True) & 0xFFFFFFFF
    k0, k1 = np.uint64(int(key[0]) & 0xFFFFFFFF), np.uint64(int(key[1]) & 0xFFFFFFFF)
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[..., 0], M1 * c[..., 2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = np.stack([(hi1 ^ c[..., 1] ^ k0) & mask, lo1, (hi0 ^ c[..., 3] ^ k1) & mask, lo0], axis=-1)
        k0, k1 = (k0 + W0) & mask, (k1 + W1) & mask
    return c.astype(np.uint32)


def device_noise(kind: str, rng_seed: int, utt0: int, batch: int, n_steps: int, width: int) -> np.ndarray:
    """the noise the decode kernels draw themselves (noise_dev == NULL): (B, n_steps, width) float32.
    laplace: e = min(fma(0.9999, u24, -0.4999), 0.49999997), u24 = (word >> 8) * 2^-24 ;
    softmax: q = -log(((word >> 9) + 0.5) * 2^-23)   (compare to ~1e-6 relative: the device logf is not libm's)."""
    tag = 0x4C41504C if kind == "laplace" else 0x45585031
    b, s, e = np.meshgrid(np.arange(batch, dtype=np.uint64), np.arange(n_steps, dtype=np.uint64),
                          np.arange(width, dtype=np.uint64), indexing="ij")
    ctr = np.stack([(b + np.uint64(utt0)) & np.uint64(0xFFFFFFFF), s, e >> np.uint64(2), np.full_like(e, tag)], axis=-1)
    words = philox4x32_10(ctr, (rng_seed & 0xFFFFFFFF, (rng_seed >> 32) & 0xFFFFFFFF))
    w = np.take_along_axis(words, (e & np.uint64(3)).astype(np.int64)[..., None], axis=-1)[..., 0]
    if kind == "laplace":
        u = (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
        v = (np.float64(np.float32(0.9999)) * u.astype(np.float64) + np.float64(np.float32(-0.4999))).astype(np.float32)   # one rounding, like fmaf
        return np.minimum(v, np.float32(0.49999997))
    u = ((w >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    return (-np.log(u.astype(np.float64))).astype(np.float32)
