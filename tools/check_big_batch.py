"""dev tool: mixed-precision step against the fp32 step at a larger batch (offset arithmetic of the bf16 operand copies).
  python tools/check_big_batch.py [B [Tf]]"""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np, torch
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.runtime import train_precision
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

B, Tf = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), (int(sys.argv[2]) if len(sys.argv) > 2 else 150)
cfg = C.ref6_laplace(1, 4)
m = mc.CSWNV(**cfg.ctor_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True).items()})
m.cuda().train()
aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
T = Tf * cfg.U; Tp = T - 2 * cfg.seg + 1
g = torch.Generator().manual_seed(2)
audio = (torch.rand(B, 1, T - cfg.seg, generator=g) * 1.8 - 0.9).cuda()
tgt = (torch.rand(B, Tp, generator=g) * 1.8 - 0.9).cuda()
grads = {}
for mode in ("fp32", "bf16"):
    for p in m.parameters(): p.grad = None
    with train_precision(mode):
        res = m(aux, audio)
        loss = mc.LaplaceLoss()(res[0].reshape(B, Tp), res[1].reshape(B, Tp), tgt, log_b=res[2].reshape(B, Tp), log=False)
        loss.backward()
    torch.cuda.synchronize()
    grads[mode] = {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}
    print(mode, "loss %.6f" % loss.item(), "peak GiB %.1f" % (torch.cuda.max_memory_allocated() / 2**30), flush=True)
big = max(float(v.norm()) for v in grads["fp32"].values())
worst = max(((float((grads["bf16"][k] - v).norm()) / (float(v.norm()) + 1e-3 * big), k) for k, v in grads["fp32"].items()))
assert all(torch.isfinite(v).all() for v in grads["bf16"].values())
print("B=%d Tf=%d: worst tensor deviation bf16 vs fp32 %.3e (%s)" % (B, Tf, worst[0], worst[1]))
assert worst[0] <= 5e-2
