#!/usr/bin/env python3
"""Headline benchmark: autoregressive Laplacian decode, BASELINE.json configs[1] ("cfg2"):
CSWNV BL6 (1x6 dilated stack, 64 hidden / 128 skip, K=2), 22.05 kHz, seg=1, lpc=0, one utterance
of Tf=600 frames (66 000 samples) per GPU, synthetic conditioning features and formula weights.

One "step" = one complete pass of the hot path over that batch: frame-rate front end + the
persistent decode launch (prologue + 66 000 generation steps), inputs (features, noise, packed
weights) already resident in HBM, samples left in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--utts B] [--frames Tf]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` prices the decode kernel against the HBM roof with the
ALGORITHMIC bytes of SURVEY.md section 8(d); `cpu_baseline` times the CPU oracle (a port of the
reference's per-step op structure, oracle/cpu_ref.py) on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from shallow_wavenet_amd import config as C, dist as D, noise as NZ           # noqa: E402
from shallow_wavenet_amd.runtime import HipNet, pack_state_dict                # noqa: E402
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict              # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

# HBM bytes per decode launch from rocprofv3 PMC passes (profiles/r01_pmc_hbm_traffic.csv), keyed by
# (utterances per GPU, frames): 2 x FETCH_SIZE (gfx950 counts wide coalesced reads at half their bytes,
# MI355X_MICROARCH.md section HBM) + WRITE_SIZE, in bytes.  Measured offline: counters cannot be read
# from inside this process.  Other shapes report null.
PMC_TRAFFIC_BYTES = {(1, 600): int((2 * 1369.8 + 257.8) * 1024)}


def algorithmic_bytes_per_position(cfg: C.NetConfig, batch: int) -> float:
    """4*W_step + 4*W_inx/U + B*(4*A0/U + state_rd + state_wr + 4)  (fp32)."""
    H, S, K, L, U = cfg.H, cfg.S, cfg.K, cfg.L, cfg.U
    hin = cfg.causal_in
    w_causal = H * hin * K + H + (2 * H if cfg.wav_conv_flag else 0)
    w_layers = L * (2 * H * H * K + 2 * H + S * H + S)
    w_head = cfg.out1_chn * S + cfg.out1_chn + cfg.n_out * cfg.out1_chn + cfg.n_out
    w_step = w_causal + w_layers + w_head
    w_inx = L * (2 * H * cfg.A + 2 * H)
    state_rd = 4 * (L * K * H + K * hin)
    state_wr = 4 * (L * H + H)
    return 4 * w_step + 4 * w_inx / U + batch * (4 * cfg.A0 / U + state_rd + state_wr + 4)


def host_threads() -> int:
    """cores this process may use (cgroup/affinity aware), capped at the GPU box's 16-core share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(cfg: C.NetConfig, sd, frames: int = 50):
    """time the oracle's free-running decode (reference per-step op structure) on the host cores."""
    from oracle import cpu_ref                      # checker only: never on the measured GPU path
    P = cpu_ref.as_params(sd)
    aux = torch.from_numpy(synth_aux(cfg, 1, frames))
    n = frames * cfg.U
    g = torch.Generator().manual_seed(1)
    noise = cpu_ref.laplace_noise(cfg, n // cfg.seg, 1, generator=g)
    best = None
    ncores = host_threads()
    for threads in sorted({1, min(ncores, 8)}):
        torch.set_num_threads(threads)
        t0 = time.time()
        cpu_ref.laplace_generate(cfg, P, aux, [n], noise)
        dt = time.time() - t0
        rate = n / dt
        if best is None or rate > best[0]:
            best = (rate, threads)
    torch.set_num_threads(ncores)
    return {"value": round(best[0], 1), "unit": "samples/s", "cores": best[1], "kind": "port",
            "sample": f"cfg2 B=1 Tf={frames} ({n} generated samples) free-running oracle decode"}


def stack_leg(cfg: C.NetConfig, net: HipNet, dev, batch: int = 64, frames: int = 150, reps: int = 10):
    """teacher-forced bf16 stack (cfg4 family, the home of the fused residual-block kernel) priced against the
    HBM roof: algorithmic bytes per position = h0 write 2H + L x (read h, write h') 4H + head reads 2LH +
    raw output 4*n_out + audio 4, conditioning 4*2H*L/U; HIP events on the launch stream."""
    T = frames * cfg.U
    aux = torch.from_numpy(synth_aux(cfg, batch, frames, seed=7)).to(dev)
    audio = (torch.rand(batch, T - cfg.seg, device=dev) * 2 - 1) * 0.5
    cond = net.frontend(aux)
    for _ in range(2):
        net.forward_bf16(aux, audio, cond=cond)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        net.forward_bf16(aux, audio, cond=cond)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    H, L = cfg.H, cfg.L
    bpp = 2 * H + L * 4 * H + 2 * L * H + 4 * cfg.n_out + 4 + 4 * 2 * H * L / cfg.U
    pos = batch * T
    ach = bpp * pos / (ms * 1e-3) / 1e9
    return {"workload": f"teacher-forced bf16 forward, BL6, {batch} x {T} positions (input + {L} fused residual blocks + head)",
            "ms": round(ms, 4), "positions_per_s": round(pos / (ms * 1e-3), 1), "bound": "hbm",
            "algorithmic_bytes_per_position": round(bpp, 1), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "note": "per-kernel split: profiles/r01_forward_b64_kernel_stats.csv (bf16_layer_units_kernel 46 % of the roof)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts", type=int, default=1, help="utterances per GPU (cfg2: 1; cfg5: 64)")
    ap.add_argument("--frames", type=int, default=600, help="conditioning frames per utterance")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, world, local = D.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

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
        "metric": "decoded samples/sec (22.05 kHz Laplacian AR decode, whole job)",
        "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
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
                     "kernel": "decode_bl6_kernel", "kernel_ms": round(kern_ms, 3),
                     "algorithmic_bytes_per_position": round(bytes_pos, 1), "positions_per_launch": positions,
                     "note": "latency-bound sequential chain; working set is L2/LDS/VGPR resident (SURVEY 7.3)"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # auxiliary legs: a failure there must not cost the headline line
        try:
            line["stack"] = stack_leg(cfg, net, dev)
        except Exception as e:                                  # noqa: BLE001
            line["stack"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            line["cpu_baseline"] = cpu_baseline(cfg, sd)
        except Exception as e:                                  # noqa: BLE001
            line["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": 0, "kind": "port",
                                    "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
