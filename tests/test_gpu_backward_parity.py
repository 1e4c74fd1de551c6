"""GPU: backward of the teacher-forced stack (HIP kernels behind torch.autograd) against the gradients the
reference itself produced (fixtures g0_*: loss.backward() through the imported reference modules).

Tolerance: per-parameter max abs error <= 2e-5 + 2e-4 * max|grad| (fp32 sums in a different order, float
atomics), digests (sum, abs-sum) of large tensors to 5e-4 relative."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.nets import dswnv as md
from shallow_wavenet_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu

LAP = [n for n in golden_names() if "_lap_" in n and not n.startswith(("g5_", "g6_", "g9_")) and "loss" in load_golden(n)[1]]
SMX = [n for n in golden_names() if "softmax" in n and not n.startswith(("g5_", "g6_", "g9_")) and "loss" in load_golden(n)[1]]


def _check(name, model, d):
    for k, p in model.named_parameters():
        g = p.grad
        assert g is not None, k
        g = g.detach().cpu().numpy().astype(np.float64)
        dig = d[f"gdig_{k}"]
        scale = max(1e-3, dig[1])
        assert abs(g.sum() - dig[0]) <= 5e-4 * scale, (name, k, g.sum(), dig[0])
        assert abs(np.abs(g).sum() - dig[1]) <= 5e-4 * scale, (name, k)
        if f"grad_{k}" in d:
            ref = d[f"grad_{k}"]
            assert np.abs(g - ref).max() <= 2e-5 + 2e-4 * np.abs(ref).max(), (name, k, np.abs(g - ref).max())


@pytest.mark.parametrize("name", LAP)
def test_laplace_gradients_match_reference(gpu_ok, name):
    cfg, d = load_golden(name)
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"])).items()})
    m.cuda().train()
    res = m(torch.from_numpy(d["aux"]).cuda(), torch.from_numpy(d["fwd_audio"]).cuda(), do=False, clip=False)
    mu, b, log_b = res[0], res[1], res[2]
    tgt = torch.from_numpy(d["loss_target"]).cuda()
    loss = mc.LaplaceLoss()(mu, b, tgt, log_b=log_b, log=False)
    if cfg.lpc > 0:
        loss = loss + 0.1 * res[3].pow(2).mean()
    assert abs(loss.item() - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    loss.backward()
    _check(name, m, d)


@pytest.mark.parametrize("name", SMX)
def test_softmax_gradients_match_reference(gpu_ok, name):
    cfg, d = load_golden(name)
    m = md.DSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"])).items()})
    m.cuda().train()
    idx = torch.from_numpy(d["fwd_audio_idx"]).cuda()
    logits = m(md.OneHot(idx, cfg.n_quantize).transpose(1, 2), torch.from_numpy(d["aux"]).cuda())
    tgt = torch.from_numpy(d["loss_target"]).cuda()
    loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tgt.reshape(-1))
    assert abs(loss.item() - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    loss.backward()
    _check(name, m, d)


def test_one_adam_step_moves_the_loss_down(gpu_ok):
    """end-to-end training step on the drop-in module: Adam(lr) as train_cswnv...py:365,872-874."""
    cfg, d = load_golden("g0_tiny_lap_s1l0_xavier")
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"])).items()})
    m.cuda().train()
    for p in m.scale_in.parameters():
        p.requires_grad = False
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    aux, audio, tgt = (torch.from_numpy(d[k]).cuda() for k in ("aux", "fwd_audio", "loss_target"))
    losses = []
    for _ in range(5):
        mu, b, log_b = m(aux, audio)
        loss = mc.LaplaceLoss()(mu, b, tgt, log_b=log_b, log=False)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
