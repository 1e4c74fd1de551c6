import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.runtime import train_precision
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict
for geom in ("bl6","ref6"):
    cfg = C.bl6_laplace(1, 0) if geom == "bl6" else C.ref6_laplace(1, 4)
    B, Tf, N = (4, 12, 20) if geom == "bl6" else (2, 8, 20)
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    T = Tf * cfg.U; Tp = T - 2 * cfg.seg + 1
    g = torch.Generator().manual_seed(2)
    audio = (torch.rand(B, 1, T - cfg.seg, generator=g) * 0.6 - 0.3).cuda()
    tgt = (torch.rand(B, Tp, generator=g) * 0.6 - 0.3).cuda()
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True).items()}
    for name, mode, fused in (("fp32", "fp32", True), ("bf16 chain", "bf16", False), ("bf16 fused", "bf16", True)):
        m = mc.CSWNV(**cfg.ctor_kwargs()); m.load_state_dict(sd); m.cuda().train()
        for p in m.scale_in.parameters(): p.requires_grad = False
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3)
        ls = []
        with train_precision(mode):
            for _ in range(N):
                m._engine().fused_backward = fused
                res = m(aux, audio)
                loss = mc.LaplaceLoss()(res[0].reshape(B, Tp), res[1].reshape(B, Tp), tgt, log_b=res[2].reshape(B, Tp), log=False)
                opt.zero_grad(set_to_none=True); loss.backward(); opt.step(); ls.append(float(loss))
        print(geom, name, round(ls[0],4), round(ls[-1],4))
