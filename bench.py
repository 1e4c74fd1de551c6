#!/usr/bin/env python3
"""Headline benchmark: autoregressive Laplacian decode, BASELINE.json configs[1] ("cfg2"):
CSWNV BL6 (1x6 dilated stack, 64 hidden / 128 skip, K=2), 22.05 kHz, seg=1, lpc=0, one utterance
of Tf=600 frames (66 000 samples) per GPU, synthetic conditioning features and formula weights.

One "step" = one complete pass of the hot path over that batch: frame-rate front end + the
persistent decode launch (prologue + 66 000 generation steps), inputs (features, noise, packed
weights) already resident in HBM, samples left in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--utts B] [--frames Tf] [--no-legs]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
         or plainly `python bench.py --gpus N`: with WORLD_SIZE unset the script starts its N ranks itself (child
         processes, before the parent makes any GPU call) and exits non-zero if any of them fails.

Prints ONE JSON line (rank 0), kept under 6 KB: every leg is numbers only (`ms`, `value`, `frac`, `bound`, `rtf`,
`us_step`); what each leg measures, its unit, dominant kernel and pricing is in profiles/LEGS.md, keyed by leg name; the
verbose per-leg dictionaries go to gpurun_out/bench_detail.json.  `roofline` prices the decode kernel against the HBM roof with the
ALGORITHMIC bytes of SURVEY.md section 8(d); `cpu_baseline` times the CPU oracle (a port of the
reference's per-step op structure, oracle/cpu_ref.py) on a bounded sample of the same workload.

At every N the line carries the `cfg5` leg (64 utterances per rank x Tf=600: BASELINE configs[4] is 8 such ranks).
At N=1 the line also carries `legs`: every other BASELINE.json config measured in the same run:
  cfg1 (softmax 16 kHz), cfg3 (seg=5, lpc=4), cfg5's 64-utterance per-GPU share, the run.sh-geometry
  (REF6) companions of cfg1/2/3/5, cfg4 (teacher-forced forward and full training step incl. Adam, BL6 and REF6,
  fp32 and bf16), and `batch_fast_generate` as a caller sees it (noise draw + launch + device->host copy).
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(argv) -> int:
    """`python bench.py --gpus N` with WORLD_SIZE unset: start the N ranks as child processes (rank r = GPU r, the
    torchrun environment contract, rendezvous on 127.0.0.1) and wait for them.  Runs before torch is imported, so
    the parent never initialises a GPU.  Rank 0 inherits stdout (the one JSON line); the other ranks' stdout goes
    to stderr.  -> 0 when every rank exited 0, otherwise the first non-zero exit code (the rest are terminated)."""
    import subprocess
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return -1                                        # nothing to launch: the caller runs as the only / a given rank
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    code = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                for q in live:                           # a failed rank leaves the others waiting in a collective
                    q.terminate()
        time.sleep(0.05)
    return code


if __name__ == "__main__":
    _rc = self_launch(sys.argv[1:])
    if _rc >= 0:
        sys.exit(_rc)

import numpy as np          # noqa: E402
import torch                # noqa: E402

from shallow_wavenet_amd import config as C, dist as D, noise as NZ           # noqa: E402
from shallow_wavenet_amd.runtime import HipNet, pack_state_dict, train_precision   # noqa: E402
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict              # noqa: E402

METRIC = "decoded samples/sec (22.05 kHz Laplacian AR decode, whole job)"
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak
MFMA_FP32_TFLOPS = 157.0         # fp32 matrix peak (exact-fp32 MFMA parity kernels)

# HBM bytes per decode launch from rocprofv3 PMC passes (profiles/r03_pmc_headline.csv: the wave-specialised kernel), keyed by
# (utterances per GPU, frames): 2 x FETCH_SIZE (gfx950 counts wide coalesced reads at half their bytes,
# MI355X_MICROARCH.md section HBM) + WRITE_SIZE, in bytes.  Measured offline: counters cannot be read
# from inside this process.  Other shapes report null.
PMC_TRAFFIC_BYTES = {(1, 600): int((2 * 1378.2 + 257.8) * 1024)}


def algorithmic_bytes_per_position(cfg: C.NetConfig, batch: int) -> float:
    """4*W_step + 4*W_inx/U + B*(4*A0/U + state_rd + state_wr + 4)  (fp32)."""
    H, S, K, L, U = cfg.H, cfg.S, cfg.K, cfg.L, cfg.U
    hin = cfg.causal_in
    if cfg.kind == "softmax" and not cfg.wav_conv_flag:
        w_causal = H * K + H                          # one-hot input: only K columns of `causal` are touched
    else:
        w_causal = H * hin * K + H + ((hin if cfg.kind == "softmax" else 1) * H + H if cfg.wav_conv_flag else 0)
    w_layers = L * (2 * H * H * K + 2 * H + S * H + S)
    w_head = cfg.out1_chn * S + cfg.out1_chn + cfg.n_out * cfg.out1_chn + cfg.n_out
    w_step = w_causal + w_layers + w_head
    w_inx = L * (2 * H * cfg.A + 2 * H)
    hin_state = H if cfg.wav_conv_flag else (1 if cfg.kind == "laplace" else 1)
    state_rd = 4 * (L * K * H + K * hin_state)
    state_wr = 4 * (L * H + H)
    return 4 * w_step + 4 * w_inx / U + batch * (4 * cfg.A0 / U + state_rd + state_wr + 4)


def stack_macs_per_position(cfg: C.NetConfig) -> int:
    """teacher-forced forward with in_x hoisted: L x (2H^2K + SH) + out_1 + out_2."""
    return cfg.L * (2 * cfg.H * cfg.H * cfg.K + cfg.S * cfg.H) + cfg.out1_chn * cfg.S + cfg.n_out * cfg.out1_chn


def host_threads() -> int:
    """cores this process may use (cgroup/affinity aware), capped at the GPU box's 16-core share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def _hip_timed(fn, reps: int, warm: int = 1, always_median: bool = False):
    """mean ms per call over `reps` back-to-back calls, HIP events on torch's current stream (the stream every launch
    helper uses).  A block that takes under 50 ms is measured three times and the median block is reported: one host
    hiccup between two launches (an allocator refill, a page fault) otherwise lands whole in a 10-call mean (seen once:
    the 0.88 ms fp32 forward leg read 5.2 ms)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()

    def block():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    t = block()
    if t < 50.0 or always_median:
        t = sorted([t, block(), block()])[1]
    return t / reps


def _hbm(bytes_per_launch: float, ms: float, **extra):
    ach = bytes_per_launch / (ms * 1e-3) / 1e9
    return dict(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                traffic=None, **extra)


def _mfma(flops_per_launch: float, ms: float, peak: float, **extra):
    ach = flops_per_launch / (ms * 1e-3) / 1e12
    return dict(bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 5),
                traffic=None, **extra)


# ------------------------------------------------------------------------------------------- cpu baseline
def cpu_baseline(cfg: C.NetConfig, sd, frames: int = 20, runs: int = 3):
    """the oracle's free-running decode (reference per-step op structure) on the host cores: >= 2 000 steady-state
    steps past the prologue, median of `runs`, for 1 thread and for the cores this process owns; the better is
    reported."""
    from oracle import cpu_ref                      # checker only: never on the measured GPU path
    P = cpu_ref.as_params(sd)
    aux = torch.from_numpy(synth_aux(cfg, 1, frames))
    n = frames * cfg.U
    g = torch.Generator().manual_seed(1)
    noise = cpu_ref.laplace_noise(cfg, n // cfg.seg, 1, generator=g)
    ncores = host_threads()
    results = {}
    for threads in sorted({1, ncores}):
        torch.set_num_threads(threads)
        rates = []
        for _ in range(runs):
            t0 = time.perf_counter()
            cpu_ref.laplace_generate(cfg, P, aux, [n], noise)
            rates.append(n / (time.perf_counter() - t0))
        results[threads] = float(np.median(rates))
    torch.set_num_threads(ncores)
    best = max(results, key=results.get)
    return {"value": round(results[best], 1), "unit": "samples/s", "cores": best, "kind": "port",
            "cores_available": ncores, "cpu_model": cpu_model(),
            "by_threads": {str(k): round(v, 1) for k, v in results.items()},
            "sample": f"cfg2 B=1 Tf={frames}: {n} generated samples after the {cfg.receptive_field}-position prologue, "
                      f"free-running oracle decode, median of {runs} runs",
            "note": "port without the reference's two growing torch.cat buffers: ~2.8x faster than the reference itself "
                    "(418 samples/s on 8 container cores, SURVEY.md 6) - the baseline is flattered, not the GPU"}


# ------------------------------------------------------------------------------------------- legs
def _net(cfg, dev, seed=1):
    sd = synth_state_dict(cfg, seed=seed, flavor="trained", identity_scale_in=True)
    return HipNet.from_state_dict(cfg, sd, dev), sd


def _module(cfg, sd, dev):
    from shallow_wavenet_amd.nets import cswnv_shift1 as mc, dswnv as md
    m = (md.DSWNV if cfg.kind == "softmax" else mc.CSWNV)(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev).eval()


def decode_leg(name: str, cfg: C.NetConfig, dev, B: int, Tf: int, fs: int, reps: int, kernel: str, variant: int = 0,
               caller: bool = False, steady: bool = False):
    """one decode workload: kernel time by HIP events (inputs resident), samples/s, real-time factor per utterance,
    HBM roofline on SURVEY 8(d)'s algorithmic bytes; optionally `batch_fast_generate` as the caller sees it."""
    soft = cfg.kind == "softmax"
    seg = 1 if soft else cfg.seg
    n_steps = Tf * cfg.U // seg
    net, sd = _net(cfg, dev)
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=11)).to(dev)
    g = torch.Generator().manual_seed(11)
    t0 = time.perf_counter()
    noise = (NZ.softmax_exponential(cfg, n_steps, B, generator=g) if soft
             else NZ.laplace_uniform(cfg, n_steps, B, generator=g))
    host_noise_s = time.perf_counter() - t0
    noise = noise.to(dev)
    cond = net.frontend(aux)
    # (the launch-chain decodes of the steady-state legs are paced by the dispatch path: a slope is the difference of two such
    #  timings, so both are medians of three blocks - one driver run read 49 us per step where every other run reads 33-38)
    ms = _hip_timed(lambda: net.decode(aux, n_steps, noise, cond=cond, variant=variant), reps, warm=1, always_median=steady)
    samples = B * n_steps * seg
    positions = n_steps + (cfg.receptive_field - seg + 1) // seg
    bpp = algorithmic_bytes_per_position(cfg, B)
    leg = {"workload": name, "utterances": B, "frames": Tf, "steps": n_steps, "samples": samples,
           "kernel_ms": round(ms, 3), "us_per_step": round(ms * 1e3 / positions, 3),
           "value": round(samples / (ms * 1e-3), 1), "unit": "samples/s",
           "real_time_factor_per_utterance": round(n_steps * seg / (ms * 1e-3) / fs, 2),
           "host_noise_draw_s": round(host_noise_s, 4),
           "roofline": _hbm(bpp * positions, ms, kernel=kernel, algorithmic_bytes_per_position=round(bpp, 1),
                            positions_per_launch=positions)}
    if steady:
        # short utterance: the rf-position prologue weighs on the figure above; the per-step cost of a long utterance is
        # the slope between this decode and one of half the length (same launch chain, same prologue)
        half = n_steps // 2
        ms_half = _hip_timed(lambda: net.decode(aux, half, noise[:, :half].contiguous(), cond=cond, variant=variant), reps, warm=1,
                             always_median=True)
        us = (ms - ms_half) * 1e3 / (n_steps - half)
        leg["steady_state_us_per_step"] = round(us, 3)
        leg["steady_state_real_time_factor"] = round(seg / (us * 1e-6) / fs, 2)
    if caller:
        m = _module(cfg, sd, dev)
        m.set_packed_engine(net)
        n_list = [n_steps * seg] * B
        seed_in = (torch.full((B, 1), cfg.n_quantize // 2, dtype=torch.int64, device=dev) if soft
                   else torch.zeros(B, seg, device=dev))
        walls = []
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.batch_fast_generate(seed_in, aux, n_list, 4410)
            walls.append(time.perf_counter() - t0)
        w = min(walls)
        leg["batch_fast_generate"] = {"wall_s": round(w, 4), "value": round(samples / w, 1), "unit": "samples/s",
                                      "real_time_factor_per_utterance": round(n_steps * seg / w / fs, 2),
                                      "includes": "front end + noise draw + decode launch + device->host copy "
                                                  "(cswnv_shift1.py:405-426 / dswnv.py:376-395)",
                                      "wall_over_kernel": round(w / (ms * 1e-3), 3)}
    return leg


def forward_leg(name: str, cfg: C.NetConfig, dev, B: int, Tf: int, mode: str, reps: int, kernel: str):
    """teacher-forced forward (cfg4 family).  BL6 is HBM-bound (algorithmic bytes per position: h0 write + L x (read h,
    write h') + head reads + raw output + audio + conditioning), REF6 MFMA-bound (2 x MAC)."""
    net, _ = _net(cfg, dev)
    soft = cfg.kind == "softmax"
    T = Tf * cfg.U
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=7)).to(dev)
    if soft:
        audio = torch.randint(0, cfg.n_quantize, (B, T - 1), device=dev)
    else:
        audio = (torch.rand(B, 1, T - cfg.seg, device=dev) * 2 - 1) * 0.5
    cond = net.frontend(aux)
    fn = (lambda: net.forward_bf16(aux, audio, cond=cond)) if mode == "bf16" else (lambda: net.forward(aux, audio, cond=cond))
    ms = _hip_timed(fn, reps, warm=2)
    pos = B * (T - 1 if soft else T - 2 * cfg.seg + 1)
    H, L = cfg.H, cfg.L
    leg = {"workload": name, "batch": B, "positions": pos, "ms": round(ms, 4), "precision": mode,
           "value": round(pos / (ms * 1e-3), 1), "unit": "positions/s"}
    flops = 2.0 * stack_macs_per_position(cfg) * pos
    leg["tflops"] = round(flops / (ms * 1e-3) / 1e12, 2)
    if cfg.H <= 64:
        eb = 2 if mode == "bf16" else 4
        bpp = eb * H + L * 2 * eb * H + eb * L * H + 4 * cfg.n_out + 4 + 4 * 2 * H * L / cfg.U
        leg["roofline"] = _hbm(bpp * pos, ms, kernel=kernel, algorithmic_bytes_per_position=round(bpp, 1))
    else:
        leg["roofline"] = _mfma(flops, ms, MFMA_BF16_TFLOPS if mode == "bf16" else MFMA_FP32_TFLOPS, kernel=kernel,
                                algorithmic_flops_per_position=2 * stack_macs_per_position(cfg))
    return leg


def train_leg(name: str, cfg: C.NetConfig, dev, B: int, Tf: int, mode: str, reps: int, kernel: str, do_prob: float = 0.0):
    """one full training step of the drop-in module as train_cswnv...py:700-874 runs it: forward, LaplaceLoss,
    backward (HIP kernels behind autograd), Adam step - the parameter re-pack that follows an optimizer step is
    inside the timed loop."""
    from shallow_wavenet_amd.nets import cswnv_shift1 as mc
    sd = synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True)
    m = mc.CSWNV(**dict(cfg.ctor_kwargs(), do_prob=do_prob))     # do_prob > 0 + forward(do=True): how run.sh trains
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.to(dev).train()
    for p in m.scale_in.parameters():
        p.requires_grad = False
    from shallow_wavenet_amd.train_driver import make_adam
    opt = make_adam([p for p in m.parameters() if p.requires_grad], 1e-4)      # as the training drivers build it (fused on a GPU)
    T = Tf * cfg.U
    Tp = T - 2 * cfg.seg + 1
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=7)).to(dev)
    audio = (torch.rand(B, 1, T - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9).to(dev)
    tgt = (torch.rand(B, Tp, generator=torch.Generator().manual_seed(3)) * 1.8 - 0.9).to(dev)
    crit = mc.LaplaceLoss()

    def step(with_opt=True):
        res = m(aux, audio, do=do_prob > 0)
        loss = crit(res[0].reshape(B, Tp), res[1].reshape(B, Tp), tgt, log_b=res[2].reshape(B, Tp), log=False)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if with_opt:
            opt.step()
        return loss

    with train_precision(mode):
        for _ in range(4):                       # the caching allocator needs a few steps to settle after the previous leg
            step()
        blocks = []                              # median of three blocks: one allocator stall inside a block of 2-5 steps
        for _ in range(3):                       # read as a 5x slower step in a driver run once
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / reps * 1e3)
        ms = sorted(blocks)[1]
        ms_nopt = _hip_timed(lambda: step(False), reps, warm=1)
    pos = B * Tp
    flops = 3 * 2.0 * stack_macs_per_position(cfg) * pos
    leg = {"workload": name, "batch": B, "positions": pos, "precision": mode, "do_prob": do_prob, "ms": round(ms, 3),
           "ms_forward_backward_only": round(ms_nopt, 3), "value": round(pos / (ms * 1e-3), 1), "unit": "positions/s",
           "tflops": round(flops / (ms * 1e-3) / 1e12, 2),
           "includes": "forward + LaplaceLoss + backward + Adam step + parameter re-pack, wall clock"}
    peak = MFMA_BF16_TFLOPS if mode == "bf16" else MFMA_FP32_TFLOPS
    leg["roofline"] = _mfma(flops, ms, peak, kernel=kernel, algorithmic_flops="3 x forward (2 x MAC)")
    if mode == "bf16" and cfg.H == 64 and cfg.K == 2 and cfg.seg == 1 and do_prob > 0:
        # fused dropout path: the plain step's streams with the hoisted conditioning replaced by sample-rate rows -
        # per position: mask 2 x 4 A0 (forward, backward) + xm16 3 x 2 A0x (written, read by two contractions) + gx16 L x 256 x 3
        # (written, read by the forward and the backward layer kernels) + d gx16 L x 256 x 3 (written, read by two contractions)
        # + d xm16 2 x 2 A0x, on top of the plain step's bytes below
        a0x = (cfg.A0 + 31) // 32 * 32
        bpp = (128 + 6 * 256 + 776 + 1800 + 6 * 1408 + 520 + 10 * 512) + 8 * cfg.A0 + 10 * a0x + 6 * cfg.L * 256
        leg["roofline_hbm"] = _hbm(float(bpp) * pos, ms, kernel="fused dropout step, all sample-rate launches", algorithmic_bytes_per_position=bpp)
    if mode == "bf16" and cfg.H == 64 and cfg.K == 2 and cfg.seg == 1 and do_prob == 0:
        # the fused BL6 path is bound by its streams, not by the matrix cores: bytes per position of the
        # sample-rate launches - forward 128 + 6 x 256 + 776, head backward 1 800, six layer launches x 1 408, the last
        # launch 520, ten weight-gradient jobs x 512
        bpp = 128 + 6 * 256 + 776 + 1800 + 6 * 1408 + 520 + 10 * 512
        leg["roofline_hbm"] = _hbm(float(bpp) * pos, ms, kernel="bl6_layer_bwd_kernel x 7 + bl6_wgrad_kernel + bl6_head_bwd_kernel"
                                   " + bf16 forward", algorithmic_bytes_per_position=bpp)
    return leg


def compact(leg: dict) -> dict:
    """numbers only (the line must stay under 6 KB; profiles/LEGS.md holds the prose, keyed by leg name):
    ms = measured time of one pass, value = the leg's rate (unit in LEGS.md), us_step / rtf for decodes (steady-state
    slope where the leg measures one), bound + frac = its roofline, frac_design = the BL6 step priced on its own streams."""
    if "error" in leg:
        return {"error": leg["error"][:120]}
    out = {"ms": leg.get("kernel_ms", leg.get("ms")), "value": leg.get("value")}
    if "us_per_step" in leg:
        out["us_step"] = leg.get("steady_state_us_per_step", leg["us_per_step"])
        out["rtf"] = leg.get("steady_state_real_time_factor", leg["real_time_factor_per_utterance"])
    if "batch_fast_generate" in leg:
        out["caller_wall_over_kernel"] = leg["batch_fast_generate"]["wall_over_kernel"]
    if "ms_forward_backward_only" in leg:
        out["ms_fwd_bwd"] = leg["ms_forward_backward_only"]
    r = leg.get("roofline")
    if r:
        out["bound"], out["frac"] = r["bound"], r["frac"]
    if "roofline_hbm" in leg:
        out["frac_design"] = leg["roofline_hbm"]["frac"]
    return out


def cfg5_leg(dev, rank: int, world: int, reps: int = 2):
    """BASELINE configs[4] as seen by this job: 64 utterances per rank x Tf = 600, CSWNV BL6 seg 1 / lpc 0, every rank
    decoding its own utterances (no data-path collective); barrier + max over ranks like the headline.
    value = samples of ALL ranks / that time."""
    cfg = C.bl6_laplace(1, 0)
    B, Tf = 64, 600
    n_steps = Tf * cfg.U
    sd = synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True)
    packed = pack_state_dict(cfg, sd) if rank == 0 else None
    net = HipNet(cfg, D.broadcast_packed(cfg, packed, dev), dev)
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=100 + rank)).to(dev)
    noise = NZ.laplace_uniform(cfg, n_steps, B, generator=torch.Generator().manual_seed(100 + rank)).to(dev)
    cond = net.frontend(aux)
    net.decode(aux, n_steps, noise, cond=cond)
    D.barrier(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        cond = net.frontend(aux)
        net.decode(aux, n_steps, noise, cond=cond)
    torch.cuda.synchronize()
    D.barrier(dev)
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev) / reps
    positions = n_steps + cfg.receptive_field
    bpp = algorithmic_bytes_per_position(cfg, B)
    return {"workload": f"cfg5: {world} rank(s) x 64 utterances x Tf=600 (66 000 samples each), CSWNV BL6 seg=1 lpc=0",
            "utterances": B * world, "ms": round(elapsed * 1e3, 3), "kernel_ms": round(elapsed * 1e3, 3),
            "us_per_step": round(elapsed * 1e6 / positions, 3),
            "value": round(world * B * n_steps / elapsed, 1), "unit": "samples/s (all ranks)",
            "real_time_factor_per_utterance": round(n_steps / elapsed / 22050.0, 2),
            "roofline": _hbm(bpp * positions, elapsed * 1e3, kernel="decode_bl6w_kernel (one workgroup per utterance)",
                             algorithmic_bytes_per_position=round(bpp, 1), positions_per_launch=positions)}


def run_legs(dev, quick: bool = False):
    legs = {}

    def add(key, fn, *a, **k):
        t0 = time.perf_counter()
        try:
            legs[key] = fn(*a, **k)
        except Exception as e:                                  # noqa: BLE001  a leg must not cost the line
            legs[key] = {"error": f"{type(e).__name__}: {e}"}
        legs[key]["leg_wall_s"] = round(time.perf_counter() - t0, 2)
        torch.cuda.empty_cache()        # a leg's cached blocks must not make the next leg's first allocations free them

    bl6 = C.bl6_laplace(1, 0)
    add("cfg2_caller", decode_leg, "cfg2: CSWNV BL6 seg=1 lpc=0, 22.05 kHz, 1 utterance x Tf=600, batch_fast_generate as called",
        bl6, dev, 1, 600, 22050, 1, "decode_bl6w_kernel", caller=True)
    add("cfg1", decode_leg, "cfg1: DSWNV BL6 softmax mu-law 256 (H=64, S=256, K=2), 16 kHz, 1 utterance x Tf=600 (48 000 steps)",
        C.bl6_softmax(), dev, 1, 600, 16000, 2, "decode_bl6_kernel<softmax>", caller=True)
    add("cfg3", decode_leg, "cfg3: CSWNV BL6 seg=5 lpc=4 (multi-sample output + LP), 22.05 kHz, 1 utterance x Tf=600 (13 200 steps)",
        C.bl6_laplace(5, 4), dev, 1, 600, 22050, 3, "decode_bl6_kernel<seg5,lpc4>", caller=True)
    # run.sh geometry (REF6: 3x2 layers, K=7, H=192/256): companions of cfg1/2/3/5, 440-step decodes (+690-position prologue)
    r_tf = 4
    stepped = "stepped decode: step_layer_kernel x L + rowvec_kernel x 2 (3 for the softmax head) + step_tail_kernel per step"
    add("ref6_cfg2", decode_leg, "REF6 companion of cfg2: CSWNV run.sh geometry (H=192, S=256, K=7, 3x2) seg=1 lpc=4, 1 utterance x Tf=4",
        C.ref6_laplace(1, 4), dev, 1, r_tf, 22050, 2, stepped, steady=True)
    add("ref6_cfg3", decode_leg, "REF6 companion of cfg3: seg=5 lpc=4, 1 utterance x Tf=4",
        C.ref6_laplace(5, 4), dev, 1, r_tf, 22050, 2, stepped, steady=True)
    add("ref6_cfg1", decode_leg, "REF6 companion of cfg1: DSWNV run.sh geometry (H=256, K=7, 3x2, Q=256), 22.05 kHz, 1 utterance x Tf=4",
        C.ref6_softmax(), dev, 1, r_tf, 22050, 2, stepped, steady=True)
    add("ref6_cfg5_share", decode_leg, "REF6 companion of cfg5's share: 64 utterances x Tf=4, seg=1 lpc=4",
        C.ref6_laplace(1, 4), dev, 64, r_tf, 22050, 1, stepped, steady=True)
    if quick:
        return legs
    # cfg4: teacher-forced stack, 8 x 16 500 (BASELINE's size) and 64 x 16 500
    units = "bf16_layer_units_kernel x 6 + bf16_input_kernel + bf16_head_kernel"
    add("cfg4_bl6_fwd_bf16", forward_leg, "cfg4 forward: BL6, 8 x 16 500 positions, bf16 MFMA stack", bl6, dev, 8, 150, "bf16", 20, units)
    add("cfg4_bl6_fwd_bf16_b64", forward_leg, "cfg4 forward at 8x the batch: BL6, 64 x 16 500 positions, bf16 MFMA stack", bl6, dev, 64, 150, "bf16", 10, units)
    add("cfg4_bl6_fwd_fp32", forward_leg, "cfg4 forward: BL6, 8 x 16 500 positions, fp32 parity kernels", bl6, dev, 8, 150, "fp32", 10,
        "tf_layer_kernel x 6 + gemm_wx_kernel x 3")
    ref6 = C.ref6_laplace(1, 4)
    add("cfg4_ref6_fwd_bf16", forward_leg, "cfg4 forward: REF6 (run.sh geometry), 8 x 16 500 positions, bf16 GEMM stack", ref6, dev, 8, 150, "bf16", 5,
        "bf16g_gemm_kernel<gate> x 6 + <relu> x 2 + <out>")
    add("cfg4_ref6_fwd_fp32", forward_leg, "cfg4 forward: REF6, 8 x 16 500 positions, fp32 parity kernels (exact-fp32 MFMA)", ref6, dev, 8, 150, "fp32", 2,
        "tf_layer_kernel x 6 + gemm_wx_kernel x 3")
    add("cfg4_bl6_step_bf16", train_leg, "cfg4 training step: BL6, 8 x 16 500, mixed precision", bl6, dev, 8, 150, "bf16", 5,
        "bf16 forward + fused backward (bl6_head_bwd / bl6_layer_bwd x 7 / bl6_wgrad) + unfold_grads")
    add("cfg4_bl6_step_fp32", train_leg, "cfg4 training step: BL6, 8 x 16 500, fp32 parity mode", bl6, dev, 8, 150, "fp32", 3,
        "tf_layer + time_gemm / reduce_gemm (exact-fp32 MFMA)")
    add("cfg4_ref6_step_bf16", train_leg, "cfg4 training step: REF6, 8 x 16 500, mixed precision", ref6, dev, 8, 150, "bf16", 3,
        "bf16g_gemm + time_gemm_bf16t / reduce_gemm_bf16s")
    add("cfg4_ref6_step_fp32", train_leg, "cfg4 training step: REF6, 8 x 16 500, fp32 parity mode", ref6, dev, 8, 150, "fp32", 2,
        "tf_layer + time_gemm / reduce_gemm (exact-fp32 MFMA)")
    add("cfg4_ref6_step_bf16_dropout", train_leg, "cfg4 training step as run.sh trains: REF6, 8 x 16 500, do_prob 0.5 (masks drawn on the "
        "device), mixed precision", ref6, dev, 8, 150, "bf16", 3,
        "time_gemm_bf16t (forward layers, in_x at sample rate) + gate_fwd / gate_bwd + reduce_gemm_bf16s", do_prob=0.5)
    add("cfg4_bl6_step_bf16_dropout", train_leg, "the same at BL6", bl6, dev, 8, 150, "bf16", 5,
        "xm16 + bf16g_gemm (in_x at sample rate) + bf16_layer_kernel<gx> x 6 + fused backward (bl6_layer_bwd<gx> x 6, "
        "bl6_wgrad incl. 24 in_x jobs) + bf16g_gemm (d xm) + xm_bwd16", do_prob=0.5)
    add("cfg4_bl6_step_bf16_b64", train_leg, "the same step at 8x the batch: BL6, 64 x 16 500, mixed precision", bl6, dev, 64, 150,
        "bf16", 3, "bf16 forward + fused backward (bl6_head_bwd / bl6_layer_bwd x 7 / bl6_wgrad) + unfold_grads")
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts", type=int, default=1, help="utterances per GPU of the headline (cfg2: 1)")
    ap.add_argument("--frames", type=int, default=600, help="conditioning frames per utterance")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--quick-legs", action="store_true", help="decode legs only")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the 64-utterance leg too (counter passes over the headline launch alone)")
    ap.add_argument("--plan-only", action="store_true",
                    help="no GPU work: join the process group (gloo), count the ranks, print the line's skeleton "
                         "(what the CPU tests drive)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where rank 0 writes the verbose per-leg dictionaries ('' = nowhere)")
    args = ap.parse_args()

    if args.plan_only:
        rank, world, local = D.init_from_env(backend="gloo")
        seen = int(round(D.sum_over_ranks(1.0, "cpu")))
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        D.barrier()
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "unit": "samples/s", "n_gpus": seen, "ranks_seen": seen,
                              "plan_only": True, "shards": [len(x) for x in D.shard_utterances(range(64 * world), world)]}),
                  flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        # reached only with WORLD_SIZE set by a launcher (self_launch handles the unset case): never print n_gpus != ranks
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}")
    rank, world, local = D.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    n_dev = torch.cuda.device_count()
    shared = world > n_dev                                       # rehearsal on a box with fewer GPUs than ranks: the ranks share
    dev = torch.device("cuda", local % n_dev)                    # the cards (dist.py then talks gloo, not RCCL)
    torch.cuda.set_device(dev)
    ranks_seen = int(round(D.sum_over_ranks(1.0, dev)))          # what the process group actually connected
    if ranks_seen != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the all-reduce saw {ranks_seen} rank(s)")

    cfg = C.bl6_laplace(seg=1, lpc=0)
    B, Tf = args.utts, args.frames
    n_steps = Tf * cfg.U // cfg.seg
    sd = synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True)
    # rank 0 packs; the flat buffer reaches the other GPUs by one RCCL broadcast over xGMI
    packed = pack_state_dict(cfg, sd) if rank == 0 else None
    net = HipNet(cfg, D.broadcast_packed(cfg, packed, dev), dev)
    # every rank decodes its own utterances (weak scaling; no data-path collective)
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=1 + rank)).to(dev)
    g = torch.Generator().manual_seed(1 + rank)
    noise = NZ.laplace_uniform(cfg, n_steps, B, generator=g).to(dev)

    def one_step():
        cond = net.frontend(aux)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out, _ = net.decode(aux, n_steps, noise, cond=cond)
        e1.record()
        return out, e0, e1

    for _ in range(args.warmup):
        one_step()
    D.barrier(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for _ in range(args.steps):
        out, e0, e1 = one_step()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    D.barrier(dev)
    torch.cuda.synchronize()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))         # decode launch, HIP events on its stream

    total_samples = world * B * n_steps * cfg.seg * args.steps
    value = total_samples / elapsed
    per_utt = value / (world * B)
    bytes_pos = algorithmic_bytes_per_position(cfg, B)
    positions = n_steps + cfg.receptive_field - cfg.seg + 1                # generation steps + prologue positions
    achieved = bytes_pos * positions / (kern_ms * 1e-3) / 1e9
    line = {
        "metric": METRIC,
        "value": round(value, 1), "unit": "samples/s", "n_gpus": ranks_seen, "ranks_seen": ranks_seen, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "cfg2: CSWNV BL6 (1x6, H=64, S=128, K=2) seg=1 lpc=0, 22.05 kHz, "
                               f"{B} utterance(s)/GPU x Tf={Tf} frames ({n_steps * cfg.seg} samples each)",
                   "utterances_per_gpu": B, "frames": Tf, "samples_per_utterance": n_steps * cfg.seg},
        "real_time_factor_per_utterance": round(per_utt / 22050.0, 2),
        "us_per_sample_step": round(kern_ms * 1e3 / positions, 3),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": PMC_TRAFFIC_BYTES.get((B, Tf)),
                     "traffic_src": "constant from an offline rocprofv3 --pmc pass (profiles/r03_pmc_headline.csv), "
                                    "not read in this process",
                     "kernel": "decode_bl6w_kernel", "kernel_ms": round(kern_ms, 3),
                     "algorithmic_bytes_per_position": round(bytes_pos, 1), "positions_per_launch": positions,
                     "note": "latency-bound sequential chain; working set on-chip"},
        "legs_doc": "profiles/LEGS.md (units, kernels, pricing per leg; frac_design = priced on the design's own stream bytes)",
    }
    if shared:
        line["rehearsal"] = f"{world} ranks share {n_dev} GPU(s) over gloo: the N > 1 code path, not a scaling measurement"
    detail = {}
    # cfg5 (64 utterances per rank) on every rank at every N: the scaling curve's second line
    if not args.no_cfg5:
        try:
            detail["cfg5"] = cfg5_leg(dev, rank, world)
        except Exception as e:                                      # noqa: BLE001
            detail["cfg5"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1:
        # auxiliary measurements: a failure there must not cost the headline line
        if not args.no_legs:
            t_legs = time.perf_counter()
            detail.update(run_legs(dev, quick=args.quick_legs))
            line["legs_wall_s"] = round(time.perf_counter() - t_legs, 1)
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg, sd)
            except Exception as e:                                  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": 0, "kind": "port",
                                        "sample": f"failed: {type(e).__name__}: {e}"}
    line["legs"] = {k: compact(v) for k, v in detail.items()}
    if rank == 0:
        if args.detail:
            try:
                os.makedirs(os.path.dirname(args.detail), exist_ok=True)
                with open(args.detail, "w") as f:
                    json.dump(dict(line, legs=detail), f, indent=1)
            except OSError:
                pass
        out_line = json.dumps(line, separators=(",", ":"))
        assert len(out_line) < 6000, len(out_line)
        print(out_line, flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
