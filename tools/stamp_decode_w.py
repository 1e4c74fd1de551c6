"""diagnostic: per-phase WORK cycles (barrier waits excluded) of the two wave groups of the wave-specialised BL6 decode kernel
   (needs the -DSWN_STAMP build: make -C shallow_wavenet_amd/csrc stamp ;
    SWN_HIP_LIB=shallow_wavenet_amd/libswn_hip_stamp.so python tools/stamp_decode_w.py)"""
import sys, os
import torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux

names = ["L0", "L1", "L2", "L3", "L4", "L5", "skip-fin", "out_1", "tail"]
cfg = C.bl6_laplace(1, 0)
sd = synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True)
net = HipNet.from_state_dict(cfg, sd, "cuda:0")
Tf = 40
n_steps = Tf * cfg.U
aux = torch.from_numpy(synth_aux(cfg, 1, Tf)).cuda()
noise = torch.empty(1, n_steps, 1).uniform_(-0.4999, 0.5).cuda()
for _ in range(2):
    out, heads = net.decode(aux, n_steps, noise, want_heads=True, variant=2)
torch.cuda.synchronize()
h = heads.flatten()[:20].cpu().numpy()
print("step total (group A clock): %.0f ticks" % h[9])
for k, n in enumerate(names):
    print("   %-10s A works %7.0f   B works %7.0f" % (n, h[k], h[10 + k]))
print("   sum        A %7.0f   B %7.0f" % (h[:9].sum(), h[10:19].sum()))
