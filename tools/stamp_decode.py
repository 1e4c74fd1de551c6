"""diagnostic: per-phase cycle shares of the BL6 decode kernel (needs the -DSWN_STAMP build:
   make -C shallow_wavenet_amd/csrc stamp ; SWN_HIP_LIB=shallow_wavenet_amd/libswn_hip_stamp.so python tools/stamp_decode.py)"""
import sys
import torch
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux

names = ["top+barrier", "L0", "L1+sk0", "L2+sk1", "L3+sk2", "L4+sk3", "L5+sk4", "sk5+fin", "out_1", "tail(out_2,sample,h0)"]
for cfg in (C.bl6_laplace(1, 0), C.bl6_laplace(5, 4), C.bl6_softmax()):
    sd = synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True)
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    Tf = 40
    soft = cfg.kind == "softmax"
    seg = 1 if soft else cfg.seg
    n_steps = Tf * cfg.U // seg
    aux = torch.from_numpy(synth_aux(cfg, 1, Tf)).cuda()
    noise = (torch.empty(1, n_steps, cfg.n_quantize).exponential_(1) if soft
             else torch.empty(1, n_steps, seg).uniform_(-0.4999, 0.5)).cuda()
    for _ in range(2):
        out, heads = net.decode(aux, n_steps, noise, want_heads=True, variant=2)
    torch.cuda.synchronize()
    h = heads.flatten()[:10].cpu().numpy()
    print(cfg.kind, "seg", seg, "cycles/step total %.0f (100 MHz ticks? see guide: s_memtime = shader clock)" % h.sum())
    for n, v in zip(names, h):
        print("   %-24s %8.0f  %5.1f%%" % (n, v, 100 * v / h.sum()))
