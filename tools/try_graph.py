"""dev tool: the whole BL6 training step (forward, loss, backward, Adam) captured in one HIP graph and replayed."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import torch
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.runtime import train_precision
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux

B, Tf = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 150)
train_precision("bf16")
cfg = C.bl6_laplace(1, 0)
def make():
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True).items()})
    m.cuda().train()
    for p in m.scale_in.parameters():
        p.requires_grad = False
    return m
aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
T = Tf * cfg.U
audio = (torch.rand(B, 1, T - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9).cuda()
Tp = T - 2 * cfg.seg + 1
tgt = (torch.rand(B, Tp, generator=torch.Generator().manual_seed(3)) * 1.8 - 0.9).cuda()

def run(m, opt, graph):
    def step():
        res = m(aux, audio)
        loss = mc.LaplaceLoss()(res[0].reshape(B, Tp), res[1].reshape(B, Tp), tgt, log_b=res[2].reshape(B, Tp), log=False)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    losses = []
    if not graph:
        for _ in range(3): losses.append(float(step()))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 50
        for _ in range(3): losses.append(float(step()))
        return ms, losses
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): losses.append(float(step()))
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 50
    for _ in range(3):
        g.replay(); losses.append(float(static_loss))
    return ms, losses

m0 = make(); o0 = torch.optim.Adam([p for p in m0.parameters() if p.requires_grad], lr=1e-4, capturable=True)
ms0, l0 = run(m0, o0, False)
print(f"eager : {ms0:.3f} ms/step  losses {l0}")
m1 = make(); o1 = torch.optim.Adam([p for p in m1.parameters() if p.requires_grad], lr=1e-4, capturable=True)
ms1, l1 = run(m1, o1, True)
print(f"graph : {ms1:.3f} ms/step  losses {l1}")
