"""dev tool: teacher-forced stack timing (cfg4 shape: B=8, Tf=150 -> 16 500 samples) fp32 vs bf16, plus
bf16 accuracy against the fp32 kernels and the oracle."""
import sys, numpy as np, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux

cfg = C.bl6_laplace(1, 0)
B, Tf = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 150)
sd = synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True)
net = HipNet.from_state_dict(cfg, sd, "cuda:0")
aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
T = Tf * cfg.U
audio = (torch.rand(B, 1, T - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9).cuda()
cond = net.frontend(aux)
r32, _ = net.forward(aux, audio, cond=cond)
r16 = net.forward_bf16(aux, audio, cond=cond)
torch.cuda.synchronize()
d = (r32 - r16).abs()
print("bf16 vs fp32 kernels: max abs %.4e, mean abs %.4e, ref scale %.3f" % (d.max().item(), d.mean().item(), r32.abs().max().item()))
if B * Tf <= 64:
    from oracle import cpu_ref
    P = cpu_ref.as_params(sd)
    ref, _ = cpu_ref.laplace_stack(cfg, P, aux.cpu(), audio.cpu())
    print("fp32 kernels vs oracle: %.3e ; bf16 vs oracle: %.3e" % ((r32.cpu() - ref).abs().max().item(), (r16.cpu() - ref).abs().max().item()))
Tp = T - 2 * cfg.seg + 1
for name, fn in (("fp32", lambda: net.forward(aux, audio, cond=cond)), ("bf16", lambda: net.forward_bf16(aux, audio, cond=cond))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    macs = (cfg.L * (2 * 64 * 128 + 128 * 64) + 128 * 128 + cfg.n_out * 128) * B * Tp
    print(f"{name}: {ms:.3f} ms / forward  ({B*Tp/ms/1e3:.2f} Mpos/s, {2*macs/ms/1e9:.2f} TFLOP/s)")
