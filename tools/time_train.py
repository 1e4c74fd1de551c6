"""dev tool: one training step (forward + backward HIP kernels) at the BASELINE cfg4 shape.
  python tools/time_train.py [B Tf [fp32|bf16 [bl6|ref6]]]   (third argument: arithmetic of the backward contractions)"""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import torch
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.synth import synth_state_dict, synth_aux

B, Tf = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 150)
from shallow_wavenet_amd.runtime import train_precision
MODE = sys.argv[3] if len(sys.argv) > 3 else "fp32"
train_precision(MODE)
ONLY = sys.argv[4].upper() if len(sys.argv) > 4 else None      # optional 4th argument: bl6 | ref6
WITH_OPT = len(sys.argv) > 5 and sys.argv[5] == "opt"           # optional 5th argument "opt": Adam step inside the loop
CHAIN = len(sys.argv) > 6 and sys.argv[6] == "chain"            # optional 6th argument "chain": generic per-layer backward
FUSED_ADAM = "fused_adam" in sys.argv                            # anywhere: torch.optim.Adam(fused=True)
DROP = 0.5 if "drop" in sys.argv else 0.0                        # anywhere "drop": do_prob = 0.5 and forward(do=True), as run.sh trains
SEG5 = "seg5" in sys.argv                                        # anywhere "seg5": seg = 5, lpc = 4 (run.sh's own setting)
for nm, cfg in (("BL6", C.bl6_laplace(5, 4) if SEG5 else C.bl6_laplace(1, 0)), ("REF6", C.ref6_laplace(5, 4) if SEG5 else C.ref6_laplace(1, 4))):
    if ONLY and nm != ONLY:
        continue
    m = mc.CSWNV(**dict(cfg.ctor_kwargs(), do_prob=DROP))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True).items()})
    m.cuda().train()
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    T = Tf * cfg.U
    audio = (torch.rand(B, 1, T - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9).cuda()
    Tp = T - 2 * cfg.seg + 1
    tgt = (torch.rand(B, Tp, generator=torch.Generator().manual_seed(3)) * 1.8 - 0.9).cuda()

    for p in m.scale_in.parameters():
        p.requires_grad = False
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4, **({"fused": True} if FUSED_ADAM else {}))

    def step():
        if CHAIN:
            m._engine().fused_backward = False
        res = m(aux, audio, do=DROP > 0)
        if cfg.seg > 1:      # (B, Tp, seg) outputs: NLL of every segment position against the same target value (timing only)
            loss = mc.LaplaceLoss()(res[0], res[1], tgt[..., None].expand_as(res[0]), log_b=res[2], log=False)
            if cfg.lpc > 0:
                loss = loss + 0.1 * res[3].pow(2).mean()
        else:
            mu, b, log_b = res[0].reshape(B, Tp), res[1].reshape(B, Tp), res[2].reshape(B, Tp)
            loss = mc.LaplaceLoss()(mu, b, tgt, log_b=log_b, log=False)
        for p in m.parameters():
            p.grad = None
        loss.backward()
        if WITH_OPT:
            opt.step()
        return loss

    for _ in range(2): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    e0.record()
    n = 5
    h0 = time.perf_counter()
    for _ in range(n): step()
    host_ms = (time.perf_counter() - h0) * 1e3 / n      # time the host needs to ISSUE a step (no sync inside)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    macs_fwd = (cfg.L * (2 * cfg.H * cfg.H * cfg.K + cfg.S * cfg.H) + cfg.S * cfg.S + cfg.n_out * cfg.S) * B * Tp
    print(f"{nm} B={B} Tf={Tf} [{MODE}]{' +Adam' if WITH_OPT else ''}{' chain' if CHAIN else ''}{' dropout 0.5' if DROP else ''}: fwd+bwd {ms:.2f} ms/step, {B*Tp/ms/1e3:.2f} Mpos/s, ~{2*3*macs_fwd/ms/1e9:.1f} TFLOP/s (3x fwd flops); host issue {host_ms:.2f} ms/step")
