"""quick timing of the decode kernels (dev tool, not the bench contract)."""
import sys, time
import numpy as np, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux

def run(cfg, B, Tf, variants=(1, 2), reps=3):
    sd = synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True)
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    soft = cfg.kind == "softmax"
    seg = 1 if soft else cfg.seg
    n_steps = Tf * cfg.U // seg
    if soft:
        noise = torch.empty(B, n_steps, cfg.n_quantize).exponential_(1).cuda()
    else:
        noise = torch.empty(B, n_steps, seg).uniform_(-0.4999, 0.5).cuda()
    cond = net.frontend(aux)
    outs = {}
    for v in variants:
        ts = []
        for r in range(reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out, _ = net.decode(aux, n_steps, noise, variant=v, cond=cond)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = min(ts)
        outs[v] = out.cpu().numpy()
        print(f"{cfg.kind} seg={seg} lpc={cfg.lpc} B={B} Tf={Tf} variant={v}: {t:.1f} ms, "
              f"{t*1e3/n_steps:.2f} us/step, {B*n_steps*seg/t*1e3/1e6:.3f} Msamples/s total, "
              f"RTF/utt={(n_steps*seg/(t/1e3))/ (16000 if soft else 22050):.1f}x", flush=True)
    if len(outs) == 2:
        a, b = outs[variants[0]], outs[variants[1]]
        if soft: print("   variants agree:", float((a == b).mean()))
        else: print("   max |diff| between variants:", float(np.abs(a - b).max()))

if __name__ == "__main__":  # noqa
    Tf = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    run(C.bl6_laplace(1, 0), 1, Tf)
    run(C.bl6_laplace(1, 0), 64, Tf)
    run(C.bl6_laplace(5, 4), 1, Tf)
    run(C.bl6_softmax(), 1, Tf)
