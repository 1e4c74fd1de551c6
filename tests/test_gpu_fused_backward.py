"""GPU: the fused per-layer backward of the BL6 class (swn_backward_bf16, csrc/swn_bwd_bl6.hip) against the generic chain
of csrc/swn_train.hip in the same mixed-precision mode, on the packed-gradient buffer.

Both round the same operands to bf16 (the hidden states are bf16 to begin with, da is rounded where it enters the
matrix cores), so what differs is the summation order and the transcendental unit (exp2/rcp against expf/tanhf):
per packed section ||g_fused - g_chain|| <= 1e-2 ||g_chain|| (measured ~1e-3).  The fp32 mode of the generic chain is the
looser yardstick of tests/test_gpu_train_bf16.py (5e-2), which now runs through the fused path too."""
import numpy as np
import pytest
import torch

from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet, layout_offsets, train_precision
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

pytestmark = pytest.mark.gpu


def _sections(cfg, gp):
    y = layout_offsets(cfg)
    names = sorted((k for k in y if k != "total"), key=lambda k: y[k])
    offs = [y[k] for k in names] + [y["total"]]
    out = {}
    for i, k in enumerate(names):
        if offs[i + 1] > offs[i]:
            out[k] = gp[offs[i]:offs[i + 1]].double().cpu().numpy()
    return out


def _run(cfg, B, Tf, seed=5):
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=3, flavor="trained", identity_scale_in=True), "cuda:0")
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    T = Tf * cfg.U
    g = torch.Generator().manual_seed(seed)
    audio = (torch.rand(B, 1, T - cfg.seg, generator=g) * 1.8 - 0.9).cuda()
    Tp = T - 2 * cfg.seg + 1
    grad_raw = (torch.randn(B, cfg.n_out, Tp, generator=g) / Tp).cuda()
    with train_precision("bf16"):
        raw, saved = net.forward_train(aux, audio)
        assert saved.get("work_bf16") is not None, "bf16 forward did not engage"
        net.fused_backward = True
        g1 = net.backward(saved, grad_raw)
        net.fused_backward = False
        g0 = net.backward(saved, grad_raw)
    torch.cuda.synchronize()
    assert not torch.equal(g1, g0), "the fused path did not engage"
    return net, g1, g0


@pytest.mark.parametrize("lpc", [0, 2])
@pytest.mark.parametrize("B,Tf", [(1, 2), (3, 12), (2, 33), (8, 150), (64, 150)])   # the last two: BASELINE cfg4 and 8x its batch
def test_fused_layers_match_the_chain(gpu_ok, B, Tf, lpc):
    cfg = C.bl6_laplace(1, lpc)
    net, g1, g0 = _run(cfg, B, Tf)
    from shallow_wavenet_amd import ops
    assert ops.backward_bf16_supported(net.dlist, B, Tf)
    assert torch.isfinite(g1).all() and torch.isfinite(g0).all()
    s1, s0 = _sections(cfg, g1), _sections(cfg, g0)
    big = max(np.linalg.norm(v) for v in s0.values())
    for k, r in s0.items():
        nr = np.linalg.norm(r)
        if nr == 0.0:
            assert np.linalg.norm(s1[k]) == 0.0, k
            continue
        err = np.linalg.norm(s1[k] - r)
        assert err <= 1e-2 * nr + 1e-4 * big, (k, err, nr)


def test_fused_backward_other_upsampling(gpu_ok):
    """U = 80 (5 chunks per frame, none ragged), U = 37 (3 chunks, ragged), and the two ends of the covered range: 16 and 112."""
    import dataclasses
    for U in (80, 37, 16, 112):
        cfg = dataclasses.replace(C.bl6_laplace(1, 0), upsampling_factor=U)
        net, g1, g0 = _run(cfg, 2, 7)
        s1, s0 = _sections(cfg, g1), _sections(cfg, g0)
        big = max(np.linalg.norm(v) for v in s0.values())
        for k, r in s0.items():
            err = np.linalg.norm(s1[k] - r)
            assert err <= 1e-2 * np.linalg.norm(r) + 1e-4 * big, (U, k, err)


def test_unsupported_geometries_report_zero(gpu_ok):
    from shallow_wavenet_amd import ops
    for cfg in (C.bl6_laplace(2, 0), C.ref6_laplace(1, 4), C.bl6_softmax()):
        assert not ops.backward_bf16_supported(ops.desc_list(cfg), 2, 8)


@pytest.mark.parametrize("U,frames,lpc", [(37, [9, 6, 4], 0), (112, [5, 3], 2)])
def test_fused_backward_against_the_oracle_under_autograd(gpu_ok, U, frames, lpc):
    """not a self-comparison: the fused path (module API, mixed-precision mode) against oracle/cpu_ref.py differentiated
    by torch autograd on the host, at two ragged shapes - U = 37 (2 whole chunks + a 5-position tail per frame, utterances
    zero-padded to different lengths) and U = 112 (the upper end of the covered range, lpc = 2).  Per parameter:
    ||g - g_ref|| <= 5e-2 ||g_ref|| + 1e-3 of the largest tensor norm (the mixed-precision yardstick of
    test_gpu_cfg4_full_size.py)."""
    import dataclasses
    from oracle import cpu_ref
    from shallow_wavenet_amd import ops
    from shallow_wavenet_amd.nets import cswnv_shift1 as mc
    cfg = dataclasses.replace(C.bl6_laplace(1, lpc), upsampling_factor=U)
    B, Tf = len(frames), max(frames)
    sd = synth_state_dict(cfg, seed=6, flavor="trained", identity_scale_in=True)
    aux_np = synth_aux(cfg, B, Tf, seed=7)
    for b, f in enumerate(frames):
        aux_np[b, :, f:] = 0.0
    aux = torch.from_numpy(aux_np)
    T = Tf * U
    Tp = T - 2 * cfg.seg + 1
    audio = torch.rand(B, 1, T - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9
    tgt = torch.rand(B, Tp, 1, generator=torch.Generator().manual_seed(3)) * 1.8 - 0.9
    if lpc == 0:
        tgt = tgt.reshape(B, Tp)

    def loss_of(res, nll):
        loss = nll(res[0], res[1], res[2])
        return loss + 0.1 * res[3].pow(2).mean() if lpc > 0 else loss

    P = cpu_ref.as_params(sd)
    for v in P.values():
        v.requires_grad_(True)
    res_r = cpu_ref.laplace_forward(cfg, P, aux, audio)
    loss_r = loss_of(res_r, lambda mu, b, lb: cpu_ref.laplace_nll(mu, b, tgt, log_b=lb))
    loss_r.backward()

    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.cuda().train()
    assert ops.backward_bf16_supported(m._engine().dlist, B, Tf)
    with train_precision("bf16"):
        res = m(aux.cuda(), audio.cuda())
        loss = loss_of(res, lambda mu, b, lb: mc.LaplaceLoss()(mu, b, tgt.cuda(), log_b=lb, log=False))
        loss.backward()
    assert m._engine().fused_backward, "the fused path is not the default"
    assert abs(loss.item() - loss_r.item()) <= 5e-3 * max(1.0, abs(loss_r.item()))
    big = max(float(np.linalg.norm(P[k].grad.numpy().ravel())) for k in P if P[k].grad is not None)
    for k, p in m.named_parameters():
        if P[k].grad is None:
            continue
        g, r = p.grad.double().cpu().numpy().ravel(), P[k].grad.double().numpy().ravel()
        err = np.linalg.norm(g - r)
        assert err <= 5e-2 * np.linalg.norm(r) + 1e-3 * big, (U, k, err, np.linalg.norm(r))
