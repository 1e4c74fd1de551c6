"""dev tool: device unfold against the torch-op unfold, repeated with a poisoned allocator (looks for unwritten outputs / races)."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np, torch
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets._autograd import unfold_packed_grads, unfold_packed_grads_device
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict


def poison():
    torch.cuda.synchronize()
    sizes = [b["size"] for seg in torch.cuda.memory_snapshot() for b in seg["blocks"] if b["state"] == "inactive"]
    ts = [torch.empty(s, dtype=torch.uint8, device="cuda") for s in sorted(sizes, reverse=True)]
    for t in ts:
        t.fill_(0xFF)
    torch.cuda.synchronize()
    del ts


CASES = {"bl6_lap": C.bl6_laplace(1, 0), "bl6_lap_seg2_lpc": C.bl6_laplace(2, 4), "ref6_lap": C.ref6_laplace(1, 4),
         "bl6_softmax": C.bl6_softmax(), "tiny_lap": C.tiny("laplace", seg=2, lpc=2), "tiny_softmax": C.tiny("softmax"),
         "tiny_softmax_audio_in": C.tiny("softmax", audio_in_flag=True)}
for name, cfg in CASES.items():
    sd = synth_state_dict(cfg, seed=4, flavor="trained")
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    names = [k for k, _ in cfg.param_shapes()]
    params = [torch.from_numpy(np.ascontiguousarray(sd[k])).cuda() for k in names]
    gp = torch.randn(net.packed.numel(), generator=torch.Generator().manual_seed(9)).cuda()
    want = [not k.startswith("scale_in") for k in names]
    ref = unfold_packed_grads(cfg, gp, dict(zip(names, params)))
    worst, bad = 0.0, None
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
        poison()
        got = unfold_packed_grads_device(net, gp, params, want)
        torch.cuda.synchronize()
        for k, w, g, p in zip(names, want, got, params):
            if not w:
                continue
            r = ref[k].reshape(p.shape)
            e = (g - r).abs().max()
            err = float(e) / max(1.0, float(r.abs().max()))
            if not (err <= worst):
                worst, bad = err, (rep, k)
    print(f"{name}: worst relative deviation {worst:.2e} at {bad}", flush=True)
