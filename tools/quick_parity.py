"""dev tool: decode one BL6 fixture with the library in SWN_HIP_LIB and print the max error."""
import sys, numpy as np, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
from conftest import load_golden
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict
for name in ("g1_bl6_lap_s1l0_b1_trained", "g1_bl6_lap_s5l4_b1_trained", "g1_bl6_softmax_b1"):
    cfg, d = load_golden(name)
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"])), "cuda:0")
    if cfg.kind == "softmax":
        from oracle import cpu_ref
        g = torch.Generator().manual_seed(int(d["noise_seed"]))
        q = cpu_ref.softmax_noise(cfg, int(d["n_samples"].max()), 1, generator=g)
        out, _ = net.decode(torch.from_numpy(d["aux"]), q.shape[0], torch.from_numpy(q).permute(1, 0, 2).contiguous(), variant=2)
        print(name, "agree", float((out.cpu().numpy()[0] == d["samples_0"]).mean()))
    else:
        noise = torch.from_numpy(d["noise"]).permute(1, 0, 2).contiguous()
        out, _ = net.decode(torch.from_numpy(d["aux"]), noise.shape[1], noise, variant=2)
        print(name, "max err", float(np.abs(out.cpu().numpy()[0] - d["samples_0"]).max()))
