"""dev tool: is the stepped decode's launch chain bound by the host (issue time ~ GPU time) or by the GPU?"""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import torch
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux
cfg = C.ref6_laplace(1, 4)
for B in (1, 8, 64):
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True), "cuda:0")
    Tf = 4
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    n = Tf * cfg.U
    noise = torch.empty(B, n, 1).uniform_(-0.4999, 0.5).cuda()
    cond = net.frontend(aux)
    net.decode(aux, n, noise, cond=cond); torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter(); net.decode(aux, n, noise, cond=cond); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        nl = (686 + n) * 7 + n * 3
        print(f"B={B}: host issue {1e3*(t1-t0):.1f} ms, until GPU done {1e3*(t2-t0):.1f} ms, ~{nl} launches -> host {1e6*(t1-t0)/nl:.2f} us/launch, total {1e6*(t2-t0)/nl:.2f} us/launch", flush=True)
