"""dev tool: where the HOST time of one BL6 training step goes (issue time per phase, no synchronisation inside)."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import torch
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.runtime import train_precision
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux

B, Tf = 8, 150
train_precision("bf16")
cfg = C.bl6_laplace(1, 0)
m = mc.CSWNV(**cfg.ctor_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True).items()})
m.cuda().train()
aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
T = Tf * cfg.U
audio = (torch.rand(B, 1, T - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9).cuda()
Tp = T - 2 * cfg.seg + 1
tgt = (torch.rand(B, Tp, generator=torch.Generator().manual_seed(3)) * 1.8 - 0.9).cuda()
for p in m.scale_in.parameters():
    p.requires_grad = False
from shallow_wavenet_amd.train_driver import make_adam
opt = make_adam([p for p in m.parameters() if p.requires_grad], 1e-4)
acc = {}
from shallow_wavenet_amd.nets import _autograd as _ag
from shallow_wavenet_amd.runtime import HipNet
def _wrap(owner, name, label):
    f = getattr(owner, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); tick(label, t0); return r
    setattr(owner, name, g)
_wrap(_ag, "unfold_packed_grads_device", "  bwd.unfold")
_wrap(HipNet, "backward", "  bwd.stack")
_wrap(HipNet, "laplace_head_backward", "  bwd.head")
_wrap(HipNet, "forward_train", "  fwd.stack")
def tick(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
def step():
    t0 = time.perf_counter(); res = m(aux, audio); tick("forward", t0)
    t0 = time.perf_counter()
    loss = mc.LaplaceLoss()(res[0].reshape(B, Tp), res[1].reshape(B, Tp), tgt, log_b=res[2].reshape(B, Tp), log=False)
    tick("loss", t0)
    t0 = time.perf_counter()
    for p in m.parameters(): p.grad = None
    loss.backward(); tick("backward", t0)
    t0 = time.perf_counter(); opt.step(); tick("adam", t0)
for _ in range(3): step()
torch.cuda.synchronize(); acc.clear()
n = 10
for _ in range(n): step()
torch.cuda.synchronize()
print({k: round(v / n * 1e3, 3) for k, v in acc.items()}, "ms/step host issue")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
