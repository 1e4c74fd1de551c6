"""dev tool: random shapes through the fused BL6 backward against the generic chain (mixed-precision mode)."""
import sys, os, dataclasses, random
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np, torch
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet, layout_offsets, train_precision
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    U = random.choice([16, 17, 37, 64, 80, 110, 112]); lpc = random.choice([0, 2, 4]); B = random.randint(1, 5); Tf = random.randint(1, 40)
    cfg = dataclasses.replace(C.bl6_laplace(1, lpc), upsampling_factor=U)
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=it, flavor="trained", identity_scale_in=True), "cuda:0")
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    T = Tf * U
    Tp = T - 2 * cfg.seg + 1
    if Tp < 1:
        continue
    g = torch.Generator().manual_seed(it)
    audio = (torch.rand(B, 1, T - cfg.seg, generator=g) * 1.8 - 0.9).cuda()
    grad_raw = (torch.randn(B, cfg.n_out, Tp, generator=g) / Tp).cuda()
    with train_precision("bf16"):
        raw, saved = net.forward_train(aux, audio)
        net.fused_backward = True; g1 = net.backward(saved, grad_raw)
        net.fused_backward = False; g0 = net.backward(saved, grad_raw)
    torch.cuda.synchronize()
    assert torch.isfinite(g1).all() and not torch.equal(g1, g0)
    y = layout_offsets(cfg); names = sorted((k for k in y if k != "total"), key=lambda k: y[k]); offs = [y[k] for k in names] + [y["total"]]
    big = float(g0.double().norm())
    for i, k in enumerate(names):
        if offs[i + 1] > offs[i]:
            a, b = g1[offs[i]:offs[i + 1]].double(), g0[offs[i]:offs[i + 1]].double()
            err = float((a - b).norm()); nb = float(b.norm())
            rel = err / (nb + 1e-4 * big)
            worst = max(worst, rel)
            assert rel <= 2e-2, (it, U, lpc, B, Tf, k, err, nb)
    print(f"case {it}: U={U} lpc={lpc} B={B} Tf={Tf} ok")
print("worst relative section error %.2e" % worst)
