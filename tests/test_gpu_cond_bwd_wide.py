"""GPU: the hoisted conditioning backward (`cond_bwd_kernel`) at upsampling factors whose per-frame tile is wider than two
64-column chunks (the five-chunk instantiation; the BASELINE / run.sh factors 80 and 110 use the two-chunk one).  Every
parameter gradient of the fp32 HIP path against the CPU oracle under torch autograd, as test_gpu_cfg4_full_size does at
cfg4's size.  Tolerance 5e-3 of the tensor norm: with ~1 300 positions a single ReLU / |x| kink that falls on different sides in
the two implementations (pre-activations within 1e-7 of zero; seen with the oracle on 8 threads against 1) moves a
gradient by 2/1 298 = 1.5e-3, an indexing error in the kernel by far more; away from kinks the deviation is ~1e-5.  Reference: the conditioning path of `CSWNV.forward`, cswnv_shift1.py:246-262 (upsampling + in_x products)."""
import dataclasses

import numpy as np
import pytest
import torch

from oracle import cpu_ref
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("U,seg,lpc", [(200, 2, 0), (130, 1, 0), (256, 5, 4), (64, 1, 0)])
def test_conditioning_gradients_at_wide_upsampling_factors(gpu_ok, U, seg, lpc):
    cfg = dataclasses.replace(C.bl6_laplace(seg, lpc), upsampling_factor=U)
    B, Tf = 2, 5
    sd = synth_state_dict(cfg, seed=U, flavor="trained", identity_scale_in=True)
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=U))
    T = Tf * U
    g = torch.Generator().manual_seed(U)
    audio = torch.rand(B, 1, T - seg, generator=g) * 1.8 - 0.9

    P = cpu_ref.as_params(sd)
    for v in P.values():
        v.requires_grad_(True)
    res_r = cpu_ref.laplace_forward(cfg, P, aux, audio)
    tgt = torch.rand(*res_r[0].shape, generator=g) * 1.8 - 0.9
    loss_r = cpu_ref.laplace_nll(res_r[0], res_r[1], tgt, log_b=res_r[2])
    loss_r.backward()

    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.cuda().train()
    res = m(aux.cuda(), audio.cuda())
    loss = mc.LaplaceLoss()(res[0], res[1], tgt.cuda(), log_b=res[2], log=False)
    loss.backward()

    assert float((res[0].cpu() - res_r[0].detach()).abs().max()) <= 1e-5
    assert abs(loss.item() - loss_r.item()) <= 1e-5 * max(1.0, abs(loss_r.item()))
    for k, p in m.named_parameters():
        ref = P[k].grad
        if ref is None:
            continue
        gk, r = p.grad.double().cpu().numpy().ravel(), ref.double().numpy().ravel()
        err = np.linalg.norm(gk - r)
        assert err <= 5e-3 * np.linalg.norm(r) + 1e-6, (k, err, np.linalg.norm(r))
