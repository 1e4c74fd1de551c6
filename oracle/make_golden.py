#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  It imports the reference's own `src/nets/{cswnv_shift1,dswnv}.py` unmodified,
loads weights from the deterministic formula of `shallow_wavenet_amd.synth` (weights are
never stored), runs `forward` / `batch_fast_generate` on CPU while recording every noise
draw, and writes small .npz files holding inputs, noise and expected outputs only.

Harness-side shims (the reference files are not touched):
  * `torch.Tensor.cuda` -> identity, because cswnv_shift1.py:345,347 call .cuda()
    unconditionally and there is no GPU here (ordinary RuntimeError otherwise);
  * `torch.Tensor.uniform_` wrapped during generate to record the Laplace noise;
  * a forward hook on `out_2` records the per-step head outputs.

Usage:  python oracle/make_golden.py [--only NAME_SUBSTR] [--skip-big]
"""
from __future__ import annotations

import argparse
import dataclasses
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src/nets")
sys.dont_write_bytecode = True

import cswnv_shift1 as ref_c   # noqa: E402  (the reference itself)
import dswnv as ref_d          # noqa: E402

from shallow_wavenet_amd import config as C            # noqa: E402
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux   # noqa: E402
from oracle import cpu_ref                               # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.Tensor.cuda = lambda self, *a, **k: self          # shim, see module docstring


def build_ref(cfg: C.NetConfig, seed: int, flavor: str):
    mod = ref_d.DSWNV if cfg.kind == "softmax" else ref_c.CSWNV
    m = mod(**cfg.ctor_kwargs())
    sd = synth_state_dict(cfg, seed=seed, flavor=flavor)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), (list(ref_sd.keys()), list(sd.keys()))
    for k in sd:
        assert tuple(ref_sd[k].shape) == sd[k].shape, (k, ref_sd[k].shape, sd[k].shape)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    assert m.receptive_field == cfg.receptive_field
    assert list(m.padding) == cfg.paddings
    return m, sd


def ragged_aux(cfg, frames, seed):
    """zero-padded in raw feature space like pad_list (decode_cswnv...py:30-48,104)."""
    B, Tf = len(frames), max(frames)
    aux = synth_aux(cfg, B, Tf, seed=seed)
    for b, f in enumerate(frames):
        aux[b, :, f:] = 0.0
    return aux


def digest(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64).ravel()
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum()] + list(a[:8]) + [0.0] * max(0, 8 - a.size))


def gen_laplace(name, cfg, frames, wseed, flavor, aux_seed, noise_seed, with_forward=True,
                with_grads=False, solo_of=None):
    t0 = time.time()
    m, sd = build_ref(cfg, wseed, flavor)
    aux = ragged_aux(cfg, frames, aux_seed)
    n_samples = [f * cfg.U for f in frames]
    B = len(frames)

    rec, heads = [], []
    orig_uniform = torch.Tensor.uniform_

    def rec_uniform(self, *a, **k):
        r = orig_uniform(self, *a, **k)
        rec.append(self.detach().clone())
        return r

    hook = m.out_2.register_forward_hook(lambda mod, i, o: heads.append(o.detach()[:, :, -1].clone()))
    torch.manual_seed(noise_seed)
    torch.Tensor.uniform_ = rec_uniform
    try:
        samples = m.batch_fast_generate(torch.zeros(B, cfg.seg), torch.from_numpy(aux), n_samples, 4410)
    finally:
        torch.Tensor.uniform_ = orig_uniform
        hook.remove()
    n_steps = len(heads)
    if cfg.lpc > 0:
        noise = torch.stack(rec).reshape(n_steps, cfg.seg, B).permute(0, 2, 1).contiguous().numpy()
    else:
        noise = torch.stack(rec).reshape(n_steps, B, cfg.seg).numpy()
    # the host-side generator must reproduce the captured stream
    g = torch.Generator().manual_seed(noise_seed)
    regen = cpu_ref.laplace_noise(cfg, n_steps, B, generator=g)
    assert np.array_equal(regen, noise), "host noise order does not match the reference draw order"

    out = dict(cfg_json=np.array(repr(cfg.to_dict())), wseed=wseed, flavor=np.array(flavor),
               frames=np.array(frames), aux=aux, noise_seed=noise_seed, noise=noise,
               heads=torch.stack(heads).numpy(), n_samples=np.array(n_samples))
    for b in range(B):
        out[f"samples_{b}"] = samples[b].astype(np.float32)

    if with_forward:
        T = max(frames) * cfg.U
        rng = np.random.Generator(np.random.PCG64([aux_seed, 5]))
        audio = rng.uniform(-0.9, 0.9, size=(B, 1, T - cfg.seg)).astype(np.float32)
        for p in m.parameters():
            p.requires_grad_(with_grads)
        res = m(torch.from_numpy(aux), torch.from_numpy(audio), do=False, clip=False)
        out["fwd_audio"] = audio
        for i, r in enumerate(res):
            out[f"fwd_{i}"] = r.detach().numpy()
        resc = m(torch.from_numpy(aux), torch.from_numpy(audio), do=False, clip=True)
        out["fwd_clip_n"] = len(resc)
        if with_grads:
            mu, b, log_b = res[0], res[1], res[2]
            tgt = torch.from_numpy(rng.uniform(-0.9, 0.9, size=tuple(mu.shape)).astype(np.float32))
            loss = ref_c.LaplaceLoss()(mu, b, tgt, log_b=log_b, log=False)
            if cfg.lpc > 0:
                loss = loss + 0.1 * res[3].pow(2).mean()
            loss.backward()
            out["loss_target"] = tgt.numpy()
            out["loss"] = np.float64(loss.item())
            for k, p in m.named_parameters():
                gnp = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
                out[f"gdig_{k}"] = digest(gnp)
                if gnp.size <= 4096:
                    out[f"grad_{k}"] = gnp
    if solo_of is not None:
        # G4: utterance `solo_of` decoded alone with its own rows of the same noise
        f1 = [frames[solo_of]]
        aux1 = aux[solo_of:solo_of + 1, :, : f1[0]].copy()
        rows = noise[: int(n_samples[solo_of] / cfg.seg), solo_of:solo_of + 1]
        # the REFERENCE decodes the solo utterance, fed its own rows of the recorded noise through a replaying uniform_
        # (lpc>0: one (1,1,1) draw per sample; lpc==0: one (1,1,seg) draw per step - the order `rows` is stored in)
        queue = [torch.from_numpy(np.ascontiguousarray(r)) for r in
                 (rows.reshape(-1, 1, 1, 1) if cfg.lpc > 0 else rows.reshape(-1, 1, 1, cfg.seg))]
        cursor = [0]

        def replay_uniform(self, *a, **k):
            self.copy_(queue[cursor[0]].reshape(self.shape))
            cursor[0] += 1
            return self

        torch.Tensor.uniform_ = replay_uniform
        try:
            solo = m.batch_fast_generate(torch.zeros(1, cfg.seg), torch.from_numpy(aux1), [n_samples[solo_of]], 4410)
        finally:
            torch.Tensor.uniform_ = orig_uniform
        assert cursor[0] == len(queue)
        out["solo_index"] = solo_of
        out["solo_samples"] = solo[0].astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"[golden] {name}: steps={n_steps} B={B} {time.time() - t0:.1f}s "
          f"|s|max={max(np.abs(s).max() for s in samples):.3f}")


def grad_sample_index(size: int, n: int = 2048) -> np.ndarray:
    """the fixed element subset of a large gradient tensor that g7 fixtures store (tests re-derive it)."""
    return np.unique(np.linspace(0, size - 1, min(n, size)).astype(np.int64))


def _store_grads(out, m):
    for k, p in m.named_parameters():
        gnp = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        out[f"gdig_{k}"] = digest(gnp)
        if gnp.size <= 4096:
            out[f"grad_{k}"] = gnp
        else:
            out[f"gsamp_{k}"] = gnp.ravel()[grad_sample_index(gnp.size)]
            out[f"gnorm_{k}"] = np.float64(np.sqrt((gnp.astype(np.float64) ** 2).sum()))


def gen_teacher_forced(name, cfg, frames, wseed, flavor, aux_seed):
    """G7: the teacher-forced stack AT THE run.sh GEOMETRY through the reference itself - `CSWNV.forward` (+clip) and
    `LaplaceLoss().backward()`, or `DSWNV.forward` and cross-entropy backward - on a ragged B=2 batch a little longer
    than the receptive field.  Outputs are stored whole (Laplace) or as head / tail / every 16th position (logits);
    gradients whole up to 4096 elements, otherwise digest + norm + a fixed 2048-element subset."""
    t0 = time.time()
    m, sd = build_ref(cfg, wseed, flavor)
    aux = ragged_aux(cfg, frames, aux_seed)
    B, T = len(frames), max(frames) * cfg.U
    rng = np.random.Generator(np.random.PCG64([aux_seed, 77]))
    out = dict(cfg_json=np.array(repr(cfg.to_dict())), wseed=wseed, flavor=np.array(flavor), frames=np.array(frames), aux=aux)
    for p in m.parameters():
        p.requires_grad_(True)
    if cfg.kind == "laplace":
        audio = rng.uniform(-0.9, 0.9, size=(B, 1, T - cfg.seg)).astype(np.float32)
        res = m(torch.from_numpy(aux), torch.from_numpy(audio), do=False, clip=False)
        out["fwd_audio"] = audio
        for i, r in enumerate(res):
            out[f"fwd_{i}"] = r.detach().numpy()
        resc = m(torch.from_numpy(aux), torch.from_numpy(audio), do=False, clip=True)
        out["fwd_clip_n"] = len(resc)
        mu, b, log_b = res[0], res[1], res[2]
        tgt = torch.from_numpy(rng.uniform(-0.9, 0.9, size=tuple(mu.shape)).astype(np.float32))
        loss = ref_c.LaplaceLoss()(mu, b, tgt, log_b=log_b, log=False)
        if cfg.lpc > 0:
            loss = loss + 0.1 * res[3].pow(2).mean()
    else:
        Q = cfg.n_quantize
        idx_in = rng.integers(0, Q, size=(B, T - 1)).astype(np.int64)
        logits = m(ref_d.OneHot(torch.from_numpy(idx_in), Q).transpose(1, 2), torch.from_numpy(aux))
        ln = logits.detach().numpy()
        out["fwd_audio_idx"] = idx_in
        out["fwd_logits_dig"] = digest(ln)
        out["fwd_logits_head"], out["fwd_logits_tail"], out["fwd_logits_s16"] = ln[:, :64], ln[:, -64:], ln[:, ::16]
        tgt = torch.from_numpy(rng.integers(0, Q, size=(B, T - 1)).astype(np.int64))
        loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, Q), tgt.reshape(-1))
    loss.backward()
    out["loss_target"] = tgt.numpy()
    out["loss"] = np.float64(loss.item())
    _store_grads(out, m)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"[golden] {name}: B={B} T={T} rf={cfg.receptive_field} loss={loss.item():.5f} {time.time() - t0:.1f}s")


def gen_seeded(name, cfg, frames, wseed, flavor, aux_seed, noise_seed):
    """G8: `batch_fast_generate` of the reference with a NON-ZERO seed waveform / seed class (its `audio` argument)."""
    t0 = time.time()
    m, sd = build_ref(cfg, wseed, flavor)
    aux = ragged_aux(cfg, frames, aux_seed)
    n_samples = [f * cfg.U for f in frames]
    B = len(frames)
    rng = np.random.Generator(np.random.PCG64([aux_seed, 31]))
    out = dict(cfg_json=np.array(repr(cfg.to_dict())), wseed=wseed, flavor=np.array(flavor), frames=np.array(frames),
               aux=aux, noise_seed=noise_seed, n_samples=np.array(n_samples))
    if cfg.kind == "laplace":
        seed = rng.uniform(-0.8, 0.8, size=(B, cfg.seg)).astype(np.float32)
        rec = []
        orig_uniform = torch.Tensor.uniform_

        def rec_uniform(self, *a, **k):
            r = orig_uniform(self, *a, **k)
            rec.append(self.detach().clone())
            return r

        torch.manual_seed(noise_seed)
        torch.Tensor.uniform_ = rec_uniform
        try:
            samples = m.batch_fast_generate(torch.from_numpy(seed), torch.from_numpy(aux), n_samples, 4410)
        finally:
            torch.Tensor.uniform_ = orig_uniform
        n_steps = max(n_samples) // cfg.seg
        noise = (torch.stack(rec).reshape(n_steps, cfg.seg, B).permute(0, 2, 1) if cfg.lpc > 0
                 else torch.stack(rec).reshape(n_steps, B, cfg.seg)).contiguous().numpy()
        out["noise"] = noise
        for b in range(B):
            out[f"samples_{b}"] = samples[b].astype(np.float32)
    else:
        seed = rng.integers(0, cfg.n_quantize, size=(B, 1)).astype(np.int64)
        torch.manual_seed(noise_seed)
        samples = m.batch_fast_generate(torch.from_numpy(seed), torch.from_numpy(aux), n_samples, 4410)
        g = torch.Generator().manual_seed(noise_seed)
        out["q"] = cpu_ref.softmax_noise(cfg, max(n_samples), B, generator=g)
        for b in range(B):
            out[f"samples_{b}"] = samples[b].astype(np.int64)
    out["seed"] = seed
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"[golden] {name}: B={B} seed={seed.ravel()[:4]} {time.time() - t0:.1f}s")


def gen_softmax(name, cfg, frames, wseed, flavor, aux_seed, noise_seed, head_stride=1,
                with_forward=True, with_grads=False):
    t0 = time.time()
    m, sd = build_ref(cfg, wseed, flavor)
    aux = ragged_aux(cfg, frames, aux_seed)
    n_samples = [f * cfg.U for f in frames]
    B, Q = len(frames), cfg.n_quantize
    heads = []
    hook = m.out_2.register_forward_hook(lambda mod, i, o: heads.append(o.detach()[:, :, -1].clone()))
    torch.manual_seed(noise_seed)
    try:
        samples = m.batch_fast_generate(torch.full((B, 1), Q // 2, dtype=torch.int64),
                                        torch.from_numpy(aux), n_samples, 4410)
    finally:
        hook.remove()
    n_steps = len(heads)
    heads = torch.stack(heads).numpy()
    g = torch.Generator().manual_seed(noise_seed)
    q = cpu_ref.softmax_noise(cfg, n_steps, B, generator=g)
    # the regenerated Exp(1) stream must reproduce the reference's indices from its own logits
    idx, margin = cpu_ref.categorical_from_noise(torch.from_numpy(heads), torch.from_numpy(q))
    ref_idx = np.stack([np.pad(s, (0, n_steps - len(s))) for s in samples], 1)
    for b in range(B):
        n = n_samples[b]
        assert np.array_equal(idx.numpy()[:n, b], ref_idx[:n, b]), "Exp(1) stream mismatch"
    out = dict(cfg_json=np.array(repr(cfg.to_dict())), wseed=wseed, flavor=np.array(flavor),
               frames=np.array(frames), aux=aux, noise_seed=noise_seed,
               q_sha=np.array(hashlib.sha256(q.tobytes()).hexdigest()),
               heads=heads[::head_stride], head_stride=head_stride,
               margin=margin.numpy().astype(np.float32), n_samples=np.array(n_samples))
    if q.nbytes <= (1 << 20):
        out["q"] = q
    for b in range(B):
        out[f"samples_{b}"] = samples[b].astype(np.int64)
    if with_forward:
        T = max(frames) * cfg.U
        rng = np.random.Generator(np.random.PCG64([aux_seed, 5]))
        idx_in = rng.integers(0, Q, size=(B, T - 1)).astype(np.int64)
        oh = ref_d.OneHot(torch.from_numpy(idx_in), Q).transpose(1, 2)
        for p in m.parameters():
            p.requires_grad_(with_grads)
        logits = m(oh, torch.from_numpy(aux))
        out["fwd_audio_idx"] = idx_in
        out["fwd_logits_dig"] = digest(logits.detach().numpy())
        out["fwd_logits_head"] = logits.detach().numpy()[:, :64]
        out["fwd_logits_tail"] = logits.detach().numpy()[:, -64:]
        if with_grads:
            tgt = torch.from_numpy(rng.integers(0, Q, size=(B, T - 1)).astype(np.int64))
            loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, Q), tgt.reshape(-1))
            loss.backward()
            out["loss_target"] = tgt.numpy()
            out["loss"] = np.float64(loss.item())
            for k, p in m.named_parameters():
                gnp = p.grad.numpy()
                out[f"gdig_{k}"] = digest(gnp)
                if gnp.size <= 4096:
                    out[f"grad_{k}"] = gnp
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"[golden] {name}: steps={n_steps} B={B} {time.time() - t0:.1f}s "
          f"min margin={margin.min():.2e}")


def gen_dropout(name, cfg, frames, wseed, flavor, aux_seed, drop_seed, p, do=True, big=False):
    """training-mode forward + backward WITH dropout (model.train(), do=True, do_prob=p): the masks the reference
    draws are re-derived with shallow_wavenet_amd.noise.dropout_masks from the same seed and checked through the
    oracle against the reference's own outputs before anything is stored.  big=True (G9: the geometries that are
    actually trained, run.sh:165-198) stores gradients the way G7 does: whole up to 4096 elements, otherwise
    digest + norm + the fixed 2048-element subset; logits additionally at every 16th position."""
    from shallow_wavenet_amd import noise as swn_noise
    t0 = time.time()
    mod = ref_d.DSWNV if cfg.kind == "softmax" else ref_c.CSWNV
    m = mod(**cfg.ctor_kwargs(), do_prob=p)
    sd = synth_state_dict(cfg, seed=wseed, flavor=flavor)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.train()
    aux = ragged_aux(cfg, frames, aux_seed)
    B, T = len(frames), max(frames) * cfg.U
    rng = np.random.Generator(np.random.PCG64([aux_seed, 9]))
    out = dict(cfg_json=np.array(repr(cfg.to_dict())), wseed=wseed, flavor=np.array(flavor), frames=np.array(frames),
               aux=aux, drop_seed=drop_seed, drop_p=np.float64(p), do=np.int64(1 if do else 0))
    P = cpu_ref.as_params(sd)
    if cfg.kind == "laplace":
        audio = rng.uniform(-0.9, 0.9, size=(B, 1, T - cfg.seg)).astype(np.float32)
        torch.manual_seed(drop_seed)
        res = m(torch.from_numpy(aux), torch.from_numpy(audio), do=do, clip=False)
        torch.manual_seed(drop_seed)
        drop = swn_noise.dropout_masks(cfg, B, max(frames), p, draw_x=do)
        raw, _ = cpu_ref.laplace_stack(cfg, P, torch.from_numpy(aux), torch.from_numpy(audio), drop=drop)
        mu_o = raw.transpose(1, 2)[:, :, :cfg.seg].reshape(res[0].shape)
        err = (mu_o - res[0].detach()).abs().max().item()
        assert err < 2e-5, f"regenerated dropout masks do not reproduce the reference ({err})"
        out["fwd_audio"] = audio
        for i, r in enumerate(res):
            out[f"fwd_{i}"] = r.detach().numpy()
        mu, b, log_b = res[0], res[1], res[2]
        tgt = torch.from_numpy(rng.uniform(-0.9, 0.9, size=tuple(mu.shape)).astype(np.float32))
        loss = ref_c.LaplaceLoss()(mu, b, tgt, log_b=log_b, log=False)
        if cfg.lpc > 0:
            loss = loss + 0.1 * res[3].pow(2).mean()
    else:
        Q = cfg.n_quantize
        idx_in = rng.integers(0, Q, size=(B, T - 1)).astype(np.int64)
        oh = ref_d.OneHot(torch.from_numpy(idx_in), Q).transpose(1, 2)
        torch.manual_seed(drop_seed)
        logits = m(oh, torch.from_numpy(aux), do=True)
        torch.manual_seed(drop_seed)
        drop = swn_noise.dropout_masks(cfg, B, max(frames), p)
        raw, _ = cpu_ref.softmax_stack(cfg, P, torch.from_numpy(idx_in), torch.from_numpy(aux), drop=drop)
        err = (raw.transpose(1, 2) - logits.detach()).abs().max().item()
        assert err < 2e-4, f"regenerated dropout masks do not reproduce the reference ({err})"
        out["fwd_audio_idx"] = idx_in
        out["fwd_logits_dig"] = digest(logits.detach().numpy())
        out["fwd_logits_head"] = logits.detach().numpy()[:, :64]
        out["fwd_logits_tail"] = logits.detach().numpy()[:, -64:]
        if big:
            out["fwd_logits_s16"] = logits.detach().numpy()[:, ::16]
        tgt = torch.from_numpy(rng.integers(0, Q, size=(B, T - 1)).astype(np.int64))
        loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, Q), tgt.reshape(-1))
    loss.backward()
    out["loss_target"] = tgt.numpy()
    out["loss"] = np.float64(loss.item())
    if big:
        _store_grads(out, m)
    else:
        for k, prm in m.named_parameters():
            gnp = prm.grad.numpy() if prm.grad is not None else np.zeros(tuple(prm.shape), np.float32)
            out[f"gdig_{k}"] = digest(gnp)
            if gnp.size <= 4096:
                out[f"grad_{k}"] = gnp
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"[golden] {name}: B={B} p={p} mask check err={err:.2e} loss={loss.item():.4f} {time.time() - t0:.1f}s")


def gen_trainstep(name, cfg, n_frames, batch_size, chunk_index, wseed, data_seed, step_seed, p, n_fft_facts):
    """one stage-4 training chunk (train_cswnv...py:700-874) through the REFERENCE's CSWNV / LaplaceLoss / LSDloss on
    the CPU: dropout masks, reparameterised sample, NLL + complex-STFT L1, backward.  The chunk slicing and the loss
    assembly are the restatement in shallow_wavenet_amd/train_driver.py (the reference script itself cannot be
    imported: soundfile / h5py are absent), so this fixture pins the network + loss modules under that assembly."""
    from shallow_wavenet_amd import train_driver as T
    t0 = time.time()
    m = ref_c.CSWNV(**cfg.ctor_kwargs(), do_prob=p)
    sd = synth_state_dict(cfg, seed=wseed, flavor="trained")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.train()
    for prm in m.parameters():
        prm.requires_grad = True
    for prm in m.scale_in.parameters():
        prm.requires_grad = False
    rng = np.random.Generator(np.random.PCG64([data_seed, 21]))
    h = rng.standard_normal((n_frames, cfg.n_aux)).astype(np.float32)
    x = np.tanh(0.3 * np.convolve(rng.standard_normal(n_frames * cfg.U), np.ones(8) / 8.0, mode="same")).astype(np.float32)
    plan = T.chunk_plan(n_frames, m.receptive_field, batch_size, cfg.seg, cfg.U)
    h_bs, x_bs, h_ss, x_ss = plan[chunk_index]
    bh, bx, trg, xp, flen = T.slice_chunk(m, torch.from_numpy(x), torch.from_numpy(h), h_bs, x_bs, h_ss, x_ss)
    fft = T.fft_sizes(n_fft_facts)
    win = [torch.hann_window(n) for n in fft]
    torch.manual_seed(step_seed)
    loss, l_lap, l_lsd, l_err = T.batch_loss(m, ref_c.LaplaceLoss(), ref_c.LSDloss(), bh, bx, trg, xp, flen, h_ss, fft, win,
                                             do=True)
    loss.backward()
    out = dict(cfg_json=np.array(repr(cfg.to_dict())), wseed=wseed, x=x, h=h, plan=np.array(plan), chunk_index=chunk_index,
               batch_size=batch_size, step_seed=step_seed, drop_p=np.float64(p), n_fft_facts=n_fft_facts, feat_len=flen,
               loss=np.float64(loss.item()), loss_laplace=np.float64(l_lap.item()),
               loss_lsd=np.float64(l_lsd.item() if l_lsd is not None else np.nan), loss_err=np.float64(l_err.item()))
    for k, prm in m.named_parameters():
        gnp = prm.grad.numpy() if prm.grad is not None else np.zeros(tuple(prm.shape), np.float32)
        out[f"gdig_{k}"] = digest(gnp)
        if gnp.size <= 4096:
            out[f"grad_{k}"] = gnp
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"[golden] {name}: plan={plan} chunk={chunk_index} feat_len={flen} loss={loss.item():.5f} "
          f"nll={l_lap.item():.5f} err={l_err.item():.5f} {time.time() - t0:.1f}s")


def gen_trainstep_softmax(name, cfg, n_frames, batch_size, chunk_index, wseed, data_seed, step_seed, p):
    """one stage-7 training chunk (train_dswnv_softmax.py:549-575) through the REFERENCE's DSWNV on the CPU (one-hot
    input, dropout, cross entropy past the receptive field, backward); chunk slicing / loss assembly from
    shallow_wavenet_amd/train_softmax_driver.py."""
    from shallow_wavenet_amd import train_softmax_driver as S
    t0 = time.time()
    m = ref_d.DSWNV(**cfg.ctor_kwargs(), do_prob=p)
    sd = synth_state_dict(cfg, seed=wseed, flavor="xavier")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.train()
    for prm in m.scale_in.parameters():
        prm.requires_grad = False
    rng = np.random.Generator(np.random.PCG64([data_seed, 22]))
    h = rng.standard_normal((n_frames, cfg.n_aux)).astype(np.float32)
    xc = rng.integers(0, cfg.n_quantize, size=n_frames * cfg.U).astype(np.int64)
    plan = S.chunk_plan(n_frames, m.receptive_field, batch_size, cfg.U)
    h_bs, x_bs, h_ss, x_ss = plan[chunk_index]
    bh, bx, trg = S.slice_chunk(torch.from_numpy(xc), torch.from_numpy(h), h_bs, x_bs, h_ss, x_ss)

    class OneHotInput:          # the reference takes the one-hot tensor the script builds with OneHot (:123-131)
        receptive_field = m.receptive_field
        def __call__(self, idx, aux, do=False):
            return m(ref_d.OneHot(idx, cfg.n_quantize).transpose(1, 2), aux, do=do)

    torch.manual_seed(step_seed)
    loss = S.batch_loss(OneHotInput(), torch.nn.CrossEntropyLoss(), bh, bx, trg, h_ss, do=True)
    loss.backward()
    out = dict(cfg_json=np.array(repr(cfg.to_dict())), wseed=wseed, xc=xc, h=h, plan=np.array(plan), chunk_index=chunk_index,
               batch_size=batch_size, step_seed=step_seed, drop_p=np.float64(p), loss=np.float64(loss.item()))
    for k, prm in m.named_parameters():
        gnp = prm.grad.numpy() if prm.grad is not None else np.zeros(tuple(prm.shape), np.float32)
        out[f"gdig_{k}"] = digest(gnp)
        if gnp.size <= 4096:
            out[f"grad_{k}"] = gnp
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"[golden] {name}: plan={plan} chunk={chunk_index} loss={loss.item():.5f} {time.time() - t0:.1f}s")


def gen_numerics():
    """G3: mu-law tables, Laplace transform grid, geometry and state-dict listings."""
    out = {}
    idx = np.arange(256)
    out["mulaw_decode_256"] = ref_d.decode_mu_law(idx, 256)
    sweep = np.linspace(-1.0, 1.0, 4001)
    out["mulaw_sweep"] = sweep
    out["mulaw_encode_sweep"] = ref_d.encode_mu_law(sweep, 256)
    out["mulaw_roundtrip"] = ref_d.encode_mu_law(ref_d.decode_mu_law(idx, 256), 256)
    eps = torch.linspace(-0.4999, 0.5, 2001)[:-1]
    out["lap_eps"] = eps.numpy()
    out["lap_t"] = (-eps.sign() * torch.log1p(-2 * eps.abs())).numpy()
    oh = ref_d.OneHot(torch.tensor([[0, 255, 256, 511, -1, 128]]), 256)
    out["onehot_argmax"] = oh.argmax(-1).numpy()
    listing = []
    for nm, cfg in [("bl6_laplace", C.bl6_laplace()), ("bl6_laplace_s5l4", C.bl6_laplace(5, 4)),
                    ("bl6_softmax", C.bl6_softmax()), ("ref6_laplace", C.ref6_laplace()),
                    ("ref6_laplace_s5", C.ref6_laplace(5, 4)), ("ref6_softmax", C.ref6_softmax()),
                    ("tiny_laplace", C.tiny()), ("tiny_softmax", C.tiny("softmax", wav_conv_flag=False))]:
        mod = ref_d.DSWNV if cfg.kind == "softmax" else ref_c.CSWNV
        m = mod(**cfg.ctor_kwargs())
        keys = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        listing.append(repr(dict(name=nm, rf=m.receptive_field, padding=list(m.padding),
                                 n_params=sum(p.numel() for p in m.parameters()), keys=keys)))
    out["geometry"] = np.array(listing)
    # default-construction / initialize() RNG order: per-parameter digests after torch.manual_seed(123)
    for nm, cfg in [("tiny_lap_s2l4", C.tiny("laplace", 2, 4)), ("tiny_softmax_wav", C.tiny("softmax", wav_conv_flag=True))]:
        mod = ref_d.DSWNV if cfg.kind == "softmax" else ref_c.CSWNV
        torch.manual_seed(123)
        m = mod(**cfg.ctor_kwargs())
        out[f"init_default_{nm}"] = np.stack([digest(v.numpy())[:3] for v in m.state_dict().values()])
        m.apply(ref_d.initialize if cfg.kind == "softmax" else ref_c.initialize)
        out[f"init_xavier_{nm}"] = np.stack([digest(v.numpy())[:3] for v in m.state_dict().values()])
    # loss modules on fixed inputs
    rng = np.random.Generator(np.random.PCG64(99))
    mu = torch.from_numpy(rng.normal(size=(3, 50)).astype(np.float32))
    b = torch.from_numpy(rng.uniform(1e-8, 0.5, size=(3, 50)).astype(np.float32))
    tg = torch.from_numpy(rng.normal(size=(3, 50)).astype(np.float32))
    out["loss_mu"], out["loss_b"], out["loss_t"] = mu.numpy(), b.numpy(), tg.numpy()
    out["loss_nll"] = np.array([ref_c.LaplaceLoss()(mu, b, tg, log=False).item(),
                                ref_c.LaplaceLoss()(mu, b, tg, clip=True, log=False).item(),
                                ref_c.LaplaceLoss()(mu, b, tg, log_b=torch.log(b), clip=True, log=False).item()])
    x = torch.from_numpy(rng.uniform(0.1, 2, size=(7, 33)).astype(np.float32))
    y = torch.from_numpy(rng.uniform(0.1, 2, size=(7, 33)).astype(np.float32))
    out["lsd_x"], out["lsd_y"] = x.numpy(), y.numpy()
    out["lsd_vals"] = np.array([ref_c.LSDloss()(x, y).item(), ref_c.LSDloss()(x, y, L2=False).item(),
                                ref_c.LSDloss()(x, y, LSD=False).item(), ref_c.LSDloss()(x, y, LSD=False, L2=False).item()])
    np.savez_compressed(os.path.join(GOLD, "g3_numerics.npz"), **out)
    print("[golden] g3_numerics")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--skip-big", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(1)
    jobs = []
    # ---- G0 tiny, B=2 ragged, Tf=(8,6)
    for seg, lpc in [(1, 0), (1, 4), (5, 4), (5, 0), (2, 4)]:
        for fl in ("xavier", "trained"):
            jobs.append((f"g0_tiny_lap_s{seg}l{lpc}_{fl}", gen_laplace,
                         dict(cfg=C.tiny("laplace", seg, lpc), frames=[8, 6], wseed=11, flavor=fl,
                              aux_seed=3, noise_seed=5, with_grads=(fl == "xavier"),
                              solo_of=(1 if (seg, lpc) in ((1, 0), (5, 4)) else None))))
    jobs.append(("g0_tiny_lap_nowav_s1l0", gen_laplace,
                 dict(cfg=C.tiny("laplace", 1, 0, wav_conv_flag=False), frames=[8, 6], wseed=12,
                      flavor="trained", aux_seed=3, noise_seed=6, with_grads=True)))
    # the (seg,1) Conv2d ahead of in_x (aux_conv2d_flag, cswnv_shift1.py:196-198)
    jobs.append(("g0_tiny_lap_c2d_s5l4_xavier", gen_laplace,
                 dict(cfg=C.tiny("laplace", 5, 4, aux_conv2d_flag=True), frames=[8, 6], wseed=15,
                      flavor="xavier", aux_seed=3, noise_seed=5, with_grads=True)))
    jobs.append(("g0_tiny_lap_c2d_s2l0_trained", gen_laplace,
                 dict(cfg=C.tiny("laplace", 2, 0, aux_conv2d_flag=True), frames=[8, 6], wseed=16,
                      flavor="trained", aux_seed=3, noise_seed=6)))
    # training-mode dropout (run.sh trains with do_prob=0.5): forward + gradients
    jobs.append(("g5_drop_tiny_lap_s1l0", gen_dropout,
                 dict(cfg=C.tiny("laplace", 1, 0), frames=[6, 5], wseed=31, flavor="xavier", aux_seed=3, drop_seed=41, p=0.5)))
    jobs.append(("g5_drop_tiny_lap_s5l4", gen_dropout,
                 dict(cfg=C.tiny("laplace", 5, 4), frames=[6, 5], wseed=32, flavor="xavier", aux_seed=3, drop_seed=42, p=0.5)))
    jobs.append(("g5_drop_tiny_lap_c2d_s2l4", gen_dropout,
                 dict(cfg=C.tiny("laplace", 2, 4, aux_conv2d_flag=True), frames=[6, 6], wseed=33, flavor="xavier",
                      aux_seed=3, drop_seed=43, p=0.3)))
    # <=2-layer Laplace stack: layer 0 always goes through dcrnn_drop in training mode, with and without do (cswnv_shift1.py:220-223)
    two = dataclasses.replace(C.tiny("laplace", 1, 0), dilation_depth=2, dilation_repeat=1)
    jobs.append(("g5_drop_two_lap_do", gen_dropout,
                 dict(cfg=two, frames=[6, 5], wseed=35, flavor="xavier", aux_seed=3, drop_seed=45, p=0.5)))
    jobs.append(("g5_drop_two_lap_nodo", gen_dropout,
                 dict(cfg=dataclasses.replace(C.tiny("laplace", 2, 4), dilation_depth=1, dilation_repeat=2), frames=[6, 5],
                      wseed=36, flavor="xavier", aux_seed=3, drop_seed=46, p=0.4, do=False)))
    jobs.append(("g5_drop_tiny_softmax", gen_dropout,
                 dict(cfg=C.tiny("softmax", wav_conv_flag=False), frames=[6, 5], wseed=34, flavor="xavier", aux_seed=3,
                      drop_seed=44, p=0.5)))
    # one stage-4 training chunk: dropout + LP mean + NLL + STFT L1 through the reference modules
    jobs.append(("g6_trainstep_tiny_s5l4", gen_trainstep,
                 dict(cfg=C.tiny("laplace", 5, 4), n_frames=40, batch_size=300, chunk_index=1, wseed=51, data_seed=7,
                      step_seed=61, p=0.5, n_fft_facts=5)))
    jobs.append(("g6_trainstep_tiny_s1l0", gen_trainstep,
                 dict(cfg=C.tiny("laplace", 1, 0), n_frames=40, batch_size=300, chunk_index=0, wseed=52, data_seed=8,
                      step_seed=62, p=0.5, n_fft_facts=5)))
    jobs.append(("g6_trainstep_tiny_s1l4_tail", gen_trainstep,
                 dict(cfg=C.tiny("laplace", 1, 4), n_frames=40, batch_size=300, chunk_index=-1, wseed=53, data_seed=9,
                      step_seed=63, p=0.25, n_fft_facts=5)))
    jobs.append(("g6_trainstep_tiny_softmax", gen_trainstep_softmax,
                 dict(cfg=C.tiny("softmax", wav_conv_flag=False), n_frames=40, batch_size=300, chunk_index=1, wseed=54,
                      data_seed=10, step_seed=64, p=0.5)))
    jobs.append(("g0_tiny_softmax", gen_softmax,
                 dict(cfg=C.tiny("softmax", wav_conv_flag=False), frames=[8, 6], wseed=13,
                      flavor="xavier", aux_seed=3, noise_seed=7, with_grads=True)))
    jobs.append(("g0_tiny_softmax_audioin", gen_softmax,
                 dict(cfg=C.tiny("softmax", wav_conv_flag=False, audio_in_flag=True), frames=[8, 6], wseed=17,
                      flavor="xavier", aux_seed=3, noise_seed=9, with_grads=True)))
    jobs.append(("g0_tiny_softmax_wav", gen_softmax,
                 dict(cfg=C.tiny("softmax", wav_conv_flag=True), frames=[8, 6], wseed=14,
                      flavor="xavier", aux_seed=3, noise_seed=8, with_grads=False)))
    # ---- G8: non-zero seed waveform / seed class handed to batch_fast_generate
    jobs.append(("g8_seed_laplace_s5l4", gen_seeded,
                 dict(cfg=C.tiny("laplace", 5, 4), frames=[6, 5], wseed=61, flavor="trained", aux_seed=9, noise_seed=21)))
    jobs.append(("g8_seed_laplace_s1l4", gen_seeded,
                 dict(cfg=C.tiny("laplace", 1, 4), frames=[6, 5], wseed=62, flavor="trained", aux_seed=9, noise_seed=22)))
    jobs.append(("g8_seed_laplace_bl6_s2l4", gen_seeded,
                 dict(cfg=C.bl6_laplace(2, 4), frames=[3, 2], wseed=63, flavor="trained", aux_seed=9, noise_seed=23)))
    jobs.append(("g8_seed_smx", gen_seeded,
                 dict(cfg=C.tiny("softmax", wav_conv_flag=False), frames=[6, 5], wseed=64, flavor="xavier", aux_seed=9,
                      noise_seed=24)))
    jobs.append(("g8_seed_smx_bl6", gen_seeded,
                 dict(cfg=C.bl6_softmax(), frames=[3, 2], wseed=65, flavor="xavier", aux_seed=9, noise_seed=25)))
    # ---- G1 BL6 (BASELINE-literal)
    for fl in ("xavier", "trained"):
        jobs.append((f"g1_bl6_lap_s1l0_b1_{fl}", gen_laplace,
                     dict(cfg=C.bl6_laplace(1, 0), frames=[4], wseed=21, flavor=fl, aux_seed=4,
                          noise_seed=9, with_forward=True)))
    jobs.append(("g1_bl6_lap_s1l0_b3_trained", gen_laplace,
                 dict(cfg=C.bl6_laplace(1, 0), frames=[4, 3, 2], wseed=21, flavor="trained",
                      aux_seed=5, noise_seed=10, with_forward=False, solo_of=2)))
    jobs.append(("g1_bl6_lap_s5l4_b1_trained", gen_laplace,
                 dict(cfg=C.bl6_laplace(5, 4), frames=[4], wseed=22, flavor="trained", aux_seed=4,
                      noise_seed=11, with_forward=True)))
    jobs.append(("g1_bl6_lap_s5l4_b3_xavier", gen_laplace,
                 dict(cfg=C.bl6_laplace(5, 4), frames=[4, 3, 2], wseed=22, flavor="xavier",
                      aux_seed=5, noise_seed=12, with_forward=False)))
    jobs.append(("g1_bl6_softmax_b1", gen_softmax,
                 dict(cfg=C.bl6_softmax(), frames=[5], wseed=23, flavor="xavier", aux_seed=4,
                      noise_seed=13)))
    jobs.append(("g1_bl6_softmax_b3", gen_softmax,
                 dict(cfg=C.bl6_softmax(), frames=[5, 4, 2], wseed=23, flavor="xavier", aux_seed=5,
                      noise_seed=14, with_forward=False, head_stride=4)))
    # ---- G2 REF6 (reference-shipped shape), 2*rf+220 steps
    if not args.skip_big:
        jobs.append(("g2_ref6_lap_s1l4_b1", gen_laplace,
                     dict(cfg=C.ref6_laplace(1, 4), frames=[15], wseed=31, flavor="trained",
                          aux_seed=6, noise_seed=15, with_forward=False)))
        jobs.append(("g2_ref6_lap_s5l4_b2", gen_laplace,
                     dict(cfg=C.ref6_laplace(5, 4), frames=[15, 12], wseed=32, flavor="trained",
                          aux_seed=6, noise_seed=16, with_forward=False)))
        jobs.append(("g2_ref6_softmax_b1", gen_softmax,
                     dict(cfg=C.ref6_softmax(), frames=[14], wseed=33, flavor="xavier", aux_seed=6,
                          noise_seed=17, with_forward=False, head_stride=8)))
        # ---- G7 REF6 teacher-forced forward + backward through the reference (B=2 ragged, T = 990 > rf = 690)
        jobs.append(("g7_ref6_tf_laplace_s1l4", gen_teacher_forced,
                     dict(cfg=C.ref6_laplace(1, 4), frames=[9, 8], wseed=41, flavor="trained", aux_seed=8)))
        jobs.append(("g7_ref6_tf_laplace_s5l4", gen_teacher_forced,
                     dict(cfg=C.ref6_laplace(5, 4), frames=[9, 8], wseed=42, flavor="xavier", aux_seed=8)))
        jobs.append(("g7_ref6_tf_smx", gen_teacher_forced,
                     dict(cfg=C.ref6_softmax(), frames=[9, 7], wseed=43, flavor="xavier", aux_seed=8)))
        # ---- G9: the training mode AS run.sh TRAINS (model.train(), forward(do=True), do_prob = 0.5, run.sh:198) at the
        # geometries that are trained, through the reference with the CPU generator seeded (B = 2 ragged, ~990 positions)
        jobs.append(("g9_drop_ref6_lap_s1l4", gen_dropout,
                     dict(cfg=C.ref6_laplace(1, 4), frames=[9, 8], wseed=71, flavor="trained", aux_seed=12, drop_seed=81,
                          p=0.5, big=True)))
        jobs.append(("g9_drop_bl6_lap_s1l0", gen_dropout,
                     dict(cfg=C.bl6_laplace(1, 0), frames=[9, 7], wseed=72, flavor="trained", aux_seed=12, drop_seed=82,
                          p=0.5, big=True)))
        jobs.append(("g9_drop_ref6_smx", gen_dropout,
                     dict(cfg=C.ref6_softmax(), frames=[9, 7], wseed=73, flavor="xavier", aux_seed=12, drop_seed=83,
                          p=0.5, big=True)))
    for name, fn, kw in jobs:
        if args.only and args.only not in name:
            continue
        fn(name, **kw)
    if not args.only or "g3" in args.only:
        gen_numerics()


if __name__ == "__main__":
    main()
