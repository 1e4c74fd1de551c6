"""dev tool: teacher-forced stack at the reference run.sh geometry (REF6), cfg4 shape: fp32 parity kernels vs the bf16 GEMM stack."""
import sys, os, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux
B, Tf = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 150)
cfg = C.ref6_laplace(1, 4)
net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True), "cuda:0")
aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
T = Tf * cfg.U
audio = (torch.rand(B, 1, T - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9).cuda()
cond = net.frontend(aux)
r32, _ = net.forward(aux, audio, cond=cond)
r16 = net.forward_bf16(aux, audio, cond=cond)
d = (r32 - r16).abs()
print("bf16 vs fp32 kernels: max abs %.4e, mean abs %.4e, ref scale %.3f" % (d.max().item(), d.mean().item(), r32.abs().max().item()))
Tp = T - 2 * cfg.seg + 1
macs = (cfg.L * (2 * cfg.H * cfg.H * cfg.K + cfg.S * cfg.H) + cfg.S * cfg.S + cfg.n_out * cfg.S) * B * Tp
for name, fn in (("fp32", lambda: net.forward(aux, audio, cond=cond)), ("bf16", lambda: net.forward_bf16(aux, audio, cond=cond))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name}: {ms:.3f} ms / forward  ({B*Tp/ms/1e3:.2f} Mpos/s, {2*macs/ms/1e9:.1f} TFLOP/s)")
