"""dev tool: forward / gradient deviation from the oracle over upsampling factors."""
import sys, os, dataclasses
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np, torch
from oracle import cpu_ref
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict
CASES = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or [(64, 1, 0), (110, 1, 0), (127, 1, 0), (128, 1, 0), (129, 1, 0), (130, 1, 0), (200, 1, 0), (200, 2, 0), (256, 5, 4)]
for U, seg, lpc in CASES:
    cfg = dataclasses.replace(C.bl6_laplace(seg, lpc), upsampling_factor=U)
    B, Tf = 2, 5
    sd = synth_state_dict(cfg, seed=U, flavor="trained", identity_scale_in=True)
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=U)); T = Tf * U
    g = torch.Generator().manual_seed(U)
    audio = torch.rand(B, 1, T - seg, generator=g) * 1.8 - 0.9
    P = cpu_ref.as_params(sd)
    for v in P.values(): v.requires_grad_(True)
    rr = cpu_ref.laplace_forward(cfg, P, aux, audio)
    tgt = torch.rand(*rr[0].shape, generator=g) * 1.8 - 0.9
    lr = cpu_ref.laplace_nll(rr[0], rr[1], tgt, log_b=rr[2]); lr.backward()
    m = mc.CSWNV(**cfg.ctor_kwargs()); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.cuda().train()
    res = m(aux.cuda(), audio.cuda())
    loss = mc.LaplaceLoss()(res[0], res[1], tgt.cuda(), log_b=res[2], log=False); loss.backward()
    d = (res[0].cpu() - rr[0].detach()).abs()
    worst = max(((float(np.linalg.norm((p.grad.cpu() - P[k].grad).numpy())) / (float(P[k].grad.norm()) + 1e-9), k) for k, p in m.named_parameters() if P[k].grad is not None))
    print(f"U={U} seg={seg}: max|dmu| {float(d.max()):.2e} at {int(d.flatten().argmax())} of {d.numel()}, |mu|max {float(rr[0].abs().max()):.2f}; worst grad rel {worst[0]:.2e} {worst[1]}", flush=True)
