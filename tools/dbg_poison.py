"""dev tool: looks for reads of uninitialised scratch - every cached free block of the allocator is filled with 0xFF (NaN
as fp32 / bf16) before each step, so a kernel that reads memory it never wrote shows up as NaN or as a large deviation.
  python tools/dbg_poison.py U,seg,lpc[,geom[,mode]] ...      geom: bl6 | ref6, mode: fp32 | bf16"""
import sys, os, dataclasses
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np, torch
from oracle import cpu_ref
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.runtime import train_precision
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict


def poison():
    torch.cuda.synchronize()
    sizes = [b["size"] for seg in torch.cuda.memory_snapshot() for b in seg["blocks"] if b["state"] == "inactive"]
    ts = [torch.empty(s, dtype=torch.uint8, device="cuda") for s in sorted(sizes, reverse=True)]
    for t in ts:
        t.fill_(0xFF)
    torch.cuda.synchronize()
    del ts


for a in sys.argv[1:]:
    f = a.split(",")
    U, seg, lpc = int(f[0]), int(f[1]), int(f[2])
    geom = f[3] if len(f) > 3 else "bl6"
    mode = f[4] if len(f) > 4 else "fp32"
    seed = int(f[5]) if len(f) > 5 else U
    base = C.bl6_laplace(seg, lpc) if geom == "bl6" else C.ref6_laplace(seg, lpc)
    cfg = dataclasses.replace(base, upsampling_factor=U)
    B, Tf = 2, 5
    sd = synth_state_dict(cfg, seed=seed, flavor="trained", identity_scale_in=True)
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=seed)); T = Tf * U
    g = torch.Generator().manual_seed(seed)
    audio = torch.rand(B, 1, T - seg, generator=g) * 1.8 - 0.9
    P = cpu_ref.as_params(sd)
    for v in P.values(): v.requires_grad_(True)
    rr = cpu_ref.laplace_forward(cfg, P, aux, audio)
    tgt = torch.rand(*rr[0].shape, generator=g) * 1.8 - 0.9
    lr = cpu_ref.laplace_nll(rr[0], rr[1], tgt, log_b=rr[2]); lr.backward()
    m = mc.CSWNV(**cfg.ctor_kwargs()); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.cuda().train()
    auxd, audiod, tgtd = aux.cuda(), audio.cuda(), tgt.cuda()
    for rep in range(2):
        for p in m.parameters(): p.grad = None
        poison()
        with train_precision(mode):
            res = m(auxd, audiod)
            loss = mc.LaplaceLoss()(res[0], res[1], tgtd, log_b=res[2], log=False); loss.backward()
        d = (res[0].detach().cpu() - rr[0].detach()).abs()
        rel = sorted(((float(np.linalg.norm((p.grad.cpu() - P[k].grad).numpy())) / (float(P[k].grad.norm()) + 1e-9), k)
                      for k, p in m.named_parameters() if P[k].grad is not None), reverse=True)
        print(f"seed {seed} U={U} seg={seg} lpc={lpc} {geom} {mode} rep {rep}: max|dmu| {float(d.max()):.2e}; worst grads " +
              ", ".join(f"{k} {r:.1e}" for r, k in rel[:3]), flush=True)
