"""dev tool: stepped decode at the run.sh geometry for several batch sizes: steady-state us per generated step
(slope between a decode and one of half the length, the 686-position prologue cancels)."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import torch
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux

def slope(cfg, B, Tf=4):
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True), "cuda:0")
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    soft = cfg.kind == "softmax"
    seg = 1 if soft else cfg.seg
    n = Tf * cfg.U // seg
    noise = (torch.empty(B, n, cfg.n_quantize).exponential_(1) if soft else torch.empty(B, n, seg).uniform_(-0.4999, 0.5)).cuda()
    cond = net.frontend(aux)
    def t(steps):
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); net.decode(aux, steps, noise[:, :steps].contiguous(), cond=cond); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best
    full, half = t(n), t(n // 2)
    us = (full - half) * 1e3 / (n - n // 2)
    print(f"{cfg.kind} seg={seg} B={B}: {us:.1f} us/step steady ({full:.1f} ms for {n} steps incl. prologue), "
          f"{seg / (us * 1e-6) / 22050:.2f}x real time per utterance, {B * seg / (us * 1e-6) / 1e3:.0f} k samples/s", flush=True)

if __name__ == "__main__":
    for B in [int(x) for x in (sys.argv[1:] or ["8", "16", "64"])]:
        slope(C.ref6_laplace(1, 4), B)
    slope(C.ref6_softmax(), 64)
    slope(C.ref6_laplace(5, 4), 64)
