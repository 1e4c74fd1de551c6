"""dev tool: 300 Adam steps on one fixed synthetic batch (BL6 cfg4 chunk shape, smaller batch) in three arithmetic modes:
fp32 parity kernels, mixed precision through the generic chain, mixed precision through the fused backward.
The three loss curves must track each other."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import torch
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.runtime import train_precision
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux

cfg = C.bl6_laplace(1, 0)
B, Tf, N = 4, 40, int(sys.argv[1]) if len(sys.argv) > 1 else 300
aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
T = Tf * cfg.U
Tp = T - 2 * cfg.seg + 1
g = torch.Generator().manual_seed(2)
audio = (torch.rand(B, 1, T - cfg.seg, generator=g) * 0.6 - 0.3).cuda()
tgt = audio[:, 0, cfg.seg - 1 + 1:].contiguous() if False else (torch.rand(B, Tp, generator=g) * 0.6 - 0.3).cuda()
curves = {}
for name, mode, fused in (("fp32", "fp32", True), ("bf16 chain", "bf16", False), ("bf16 fused", "bf16", True)):
    torch.manual_seed(0)
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True).items()})
    m.cuda().train()
    for p in m.scale_in.parameters():
        p.requires_grad = False
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    losses = []
    with train_precision(mode):
        for it in range(N):
            m._engine().fused_backward = fused
            res = m(aux, audio)
            loss = mc.LaplaceLoss()(res[0].reshape(B, Tp), res[1].reshape(B, Tp), tgt, log_b=res[2].reshape(B, Tp), log=False)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            if it % 25 == 0 or it == N - 1:
                losses.append(round(float(loss), 4))
    curves[name] = losses
    print(f"{name:11s}", losses)
a, b, c = curves["fp32"], curves["bf16 chain"], curves["bf16 fused"]
print("max |fused - chain| =", max(abs(x - y) for x, y in zip(c, b)), " max |fused - fp32| =", max(abs(x - y) for x, y in zip(c, a)))
