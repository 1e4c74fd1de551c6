"""GPU: BASELINE cfg4 at its full size (teacher-forced stack, batch 8 x 16 500 samples, BL6): forward, loss and every
parameter gradient of the HIP path (fp32 parity mode, through the drop-in module and torch.autograd) against the CPU
oracle differentiated by torch autograd on the same inputs (oracle/cpu_ref.py restates CSWNV.forward op by op; it is
pinned to the reference by the g0_* gradient fixtures at sizes the reference finishes in seconds).

Tolerances: head outputs <= 1e-5 abs (north_star), loss <= 1e-5 relative; gradients are sums over 132 000 positions
accumulated in a different order (tiles, float atomics): per tensor ||g - g_ref||_2 <= 1e-3 ||g_ref||_2 + 1e-6.
The mixed-precision mode of the same step is checked against the same oracle gradients at 5e-2."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

pytestmark = pytest.mark.gpu


def test_cfg4_forward_loss_and_gradients_match_the_oracle(gpu_ok):
    cfg = C.bl6_laplace(1, 0)
    B, Tf = 8, 150
    sd = synth_state_dict(cfg, seed=4, flavor="trained", identity_scale_in=True)
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=4))
    T = Tf * cfg.U
    Tp = T - 2 * cfg.seg + 1
    audio = torch.rand(B, 1, T - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9
    tgt = torch.rand(B, Tp, generator=torch.Generator().manual_seed(3)) * 1.8 - 0.9

    # oracle on the host cores
    P = cpu_ref.as_params(sd)
    for v in P.values():
        v.requires_grad_(True)
    mu_r, b_r, logb_r = cpu_ref.laplace_forward(cfg, P, aux, audio)
    loss_r = cpu_ref.laplace_nll(mu_r, b_r, tgt, log_b=logb_r)
    loss_r.backward()

    # HIP path behind the reference's module API
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.cuda().train()
    mu, b, logb = m(aux.cuda(), audio.cuda())
    loss = mc.LaplaceLoss()(mu, b, tgt.cuda(), log_b=logb, log=False)
    loss.backward()

    assert float((mu.cpu() - mu_r.detach()).abs().max()) <= 1e-5
    assert float((b.cpu() - b_r.detach()).abs().max()) <= 1e-5
    assert abs(loss.item() - loss_r.item()) <= 1e-5 * max(1.0, abs(loss_r.item()))
    seen = 0
    for k, p in m.named_parameters():
        ref = P[k].grad
        assert ref is not None and p.grad is not None, k
        g, r = p.grad.double().cpu().numpy().ravel(), ref.double().numpy().ravel()
        err = np.linalg.norm(g - r)
        assert err <= 1e-3 * np.linalg.norm(r) + 1e-6, (k, err, np.linalg.norm(r))
        seen += 1
    assert seen == len(P)

    # the same step in the mixed-precision mode (bf16 forward feeding the backward, bf16 operands in the contractions):
    # outputs within 5e-3 of the fp32 ones, gradients within 5e-2 (tensor norm; floor for the cancelling scalar bias)
    from shallow_wavenet_amd.runtime import train_precision
    for p in m.parameters():
        p.grad = None
    with train_precision("bf16"):
        mu16, b16, logb16 = m(aux.cuda(), audio.cuda())
        loss16 = mc.LaplaceLoss()(mu16, b16, tgt.cuda(), log_b=logb16, log=False)
        loss16.backward()
    assert float((mu16.cpu() - mu_r.detach()).abs().max()) <= 5e-3 * max(1.0, float(mu_r.abs().max()))
    assert abs(loss16.item() - loss_r.item()) <= 2e-3 * max(1.0, abs(loss_r.item()))
    big = max(float(np.linalg.norm(P[k].grad.numpy().ravel())) for k, _ in m.named_parameters())
    for k, p in m.named_parameters():
        g, r = p.grad.double().cpu().numpy().ravel(), P[k].grad.double().numpy().ravel()
        err = np.linalg.norm(g - r)
        assert err <= 5e-2 * np.linalg.norm(r) + 1e-3 * big, (k, err, np.linalg.norm(r))
