"""Decode drivers (stage 5 / 8 of egs/shallow-wavenet/run.sh:651-708,824-878): the counterpart of
`src/bin/decode_cswnv_laplace-shift1.py` and `src/bin/decode_dswnv_softmax.py`.

Kept from the reference: CLI flags (decode_cswnv...py:117-148), length sort + `np.array_split`
batching (:77-84), zero `pad_list` in raw feature space (:30-48), seed waveform zeros(seg) / mu-law
class 128, all utterances of a batch run max(n_samples) steps, 16-bit PCM WAV output, the two
summary log lines (:256-259), contiguous `np.array_split` sharding over GPUs (:200-201).

`--n_gpus N` as run.sh:675-684 passes it: the parent (which never touches a GPU) starts one child process per
shard (spawn context, rank r on GPU r of `--GPU_device_str` - decode_cswnv...py:261-274) and returns non-zero if
any child does.  Different on purpose: rank 0 packs the checkpoint once and broadcasts the flat parameter buffer
over RCCL instead of every process reading the checkpoint file; the same ranks can be started by `torchrun`
(RANK/WORLD_SIZE in the environment).  `--n_gpus 1` runs in-process.

Device-drawn sampling noise (`--noise_source device`, the softmax default) is keyed by ONE value drawn after
`torch.manual_seed(--seed)` and by each utterance's position in the unsorted `--feats` list, so an utterance draws
the same stream whatever `--n_gpus` and wherever the length sort puts it.

Feature files: `<utt>.npy` (T x n_aux float arrays) always; `<utt>.h5` with the dataset named by
`config.string_path` when h5py is importable (it is not in the build image).
"""
from __future__ import annotations

import argparse
import glob
import json
import logging
import math
import os
import sys
import time
import wave
from types import SimpleNamespace
from typing import Iterator, List, Sequence, Tuple

import numpy as np
import torch

from . import artefacts, dist as D, featio
from .nets import cswnv_shift1 as laplace_mod
from .nets import dswnv as softmax_mod
from .runtime import HipNet, pack_state_dict


# --------------------------------------------------------------------------- feature / list I/O
def read_feature(path: str, string_path: str = "/feat_org_lf0") -> np.ndarray:
    """(Tf, n_aux) features of one utterance: HDF5 / .npz dataset `string_path` or a plain .npy (featio.py)."""
    return featio.read_dataset(featio.resolve(path), string_path)


def feature_frames(path: str, string_path: str) -> int:
    return int(featio.dataset_shape(featio.resolve(path), string_path)[0])


def list_features(feats: str, string_path: str = None) -> List[str]:
    """directory -> sorted recursive *.h5 (and the .npz / .npy side formats) ; file -> one path per line
    (utils.py:129-160).  An utterance present in several formats is listed once (.h5 before .npz before .npy); with
    `string_path` given, container files that do not hold that dataset (a `stats.npz` written by calc_stats next to
    the features) are not utterances and are skipped."""
    if os.path.isdir(feats):
        best = {}
        for rank, ext in enumerate((".h5", ".npz", ".npy")):
            for f in glob.glob(os.path.join(feats, "**", "*" + ext), recursive=True):
                best.setdefault(f[: -len(ext)], (rank, f))
        out = sorted(f for _, f in best.values())
        if string_path is not None:
            out = [f for f in out if f.endswith(".npy") or featio.check_dataset(f, string_path)]
        return out
    if os.path.isfile(feats):
        with open(feats) as f:
            return [ln.strip() for ln in f if ln.strip()]
    raise FileNotFoundError("--feats should be directory or list.")


def pad_list(batch_list: Sequence[np.ndarray], pad_value: float = 0.0) -> np.ndarray:
    """(T_i, C) arrays -> (B, T_max, C), zero padded in RAW feature space (before scale_in): the
    last frames of a shorter utterance therefore depend on batch composition (reference quirk)."""
    maxlen = max(b.shape[0] for b in batch_list)
    out = np.full((len(batch_list), maxlen, batch_list[0].shape[-1]), pad_value, dtype=np.float64)
    for i, b in enumerate(batch_list):
        out[i, : b.shape[0]] = b
    return out


def plan_batches(feat_list: Sequence[str], frames: Sequence[int], batch_size: int) -> List[List[str]]:
    """sort by frame count (np.argsort like the reference) and cut into ceil(N/bs) near-equal batches."""
    if len(feat_list) == 0:          # a rank with an empty shard (fewer utterances than GPUs) decodes nothing
        return []
    idx = np.argsort(list(frames))
    ordered = [feat_list[i] for i in idx]
    n_batch = math.ceil(len(ordered) / batch_size)
    return [a.tolist() for a in np.array_split(ordered, n_batch)]


def decode_batches(feat_list: Sequence[str], batch_size: int, string_path: str, upsampling_factor: int,
                   global_index: Sequence[int] = None
                   ) -> Iterator[Tuple[List[str], np.ndarray, List[int]]]:
    """yields (utterance ids, padded features, n_samples) per batch; with `global_index` (position of every file of
    `feat_list` in the unsorted full list) a fourth value: those positions for the batch's utterances."""
    frames = [feature_frames(f, string_path) for f in feat_list]
    where = {f: i for i, f in enumerate(feat_list)}
    for batch in plan_batches(feat_list, frames, batch_size):
        hs = [read_feature(f, string_path) for f in batch]
        ids = [os.path.splitext(os.path.basename(f))[0] for f in batch]
        item = (ids, pad_list(hs), [h.shape[0] * upsampling_factor for h in hs])
        yield item if global_index is None else item + ([int(global_index[where[f]]) for f in batch],)


def write_wav_pcm16(path: str, samples: np.ndarray, fs: int) -> None:
    """float [-1,1] -> 16-bit PCM exactly as libsndfile does for soundfile.write(..., "PCM_16"):
    lrint(x * 0x7FFF) after the caller's np.clip (decode_cswnv...py:253-254)."""
    pcm = np.rint(np.asarray(samples, dtype=np.float64) * 32767.0).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(fs))
        w.writeframes(pcm.tobytes())


def load_config(path: str):
    """model.conf: the pickled argparse Namespace the training stage wrote (train_cswnv...py:293; ours or the
    reference's), a dict of the same fields, or JSON - loaded without executing anything (artefacts.py)."""
    return artefacts.load_config(path)


# --------------------------------------------------------------------------- model construction
def build_model(kind: str, config):
    if kind == "laplace":
        return laplace_mod.CSWNV(
            n_aux=config.n_aux, skip_chn=config.skip_chn, hid_chn=config.hid_chn,
            dilation_depth=config.dilation_depth, dilation_repeat=config.dilation_repeat,
            kernel_size=config.kernel_size, aux_kernel_size=config.aux_kernel_size,
            aux_dilation_size=config.aux_dilation_size, seg=config.seg, lpc=config.lpc,
            aux_conv2d_flag=config.aux_conv2d_flag, wav_conv_flag=config.wav_conv_flag,
            upsampling_factor=config.upsampling_factor)
    return softmax_mod.DSWNV(
        n_quantize=config.n_quantize, n_aux=config.n_aux, hid_chn=config.hid_chn, skip_chn=config.skip_chn,
        dilation_depth=config.dilation_depth, dilation_repeat=config.dilation_repeat,
        kernel_size=config.kernel_size, aux_kernel_size=config.aux_kernel_size,
        aux_dilation_size=config.aux_dilation_size, audio_in_flag=getattr(config, "audio_in", False),
        wav_conv_flag=config.wav_conv_flag, upsampling_factor=config.upsampling_factor)


def gpu_decode(kind: str, args, config, feat_list: Sequence[str], device, packed_src_rank: int = 0,
               global_index: Sequence[int] = None, rng_key: int = None):
    """decode one shard on one GPU (decode_cswnv...py:204-259).  global_index / rng_key: every utterance's position in
    the unsorted full list and the run's one generator key - device-drawn noise then depends on neither batching nor
    sharding (module attributes noise_utterance_ids / noise_rng_seed)."""
    with torch.no_grad():
        model = build_model(kind, config)
        rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        cfg = model._cfg
        if rank == packed_src_rank:
            sd = artefacts.load_checkpoint(args.checkpoint)["model"]       # per-epoch or final file (run.sh:658)
            model.load_state_dict(sd)
            packed = pack_state_dict(cfg, model.state_dict())
        else:
            packed = None
        model.to(device)
        model.eval()
        model.noise_source = getattr(args, "noise_source", None)
        model.noise_rng_seed = rng_key
        if global_index is None:
            global_index = list(range(len(feat_list)))
        model.set_packed_engine(HipNet(cfg, D.broadcast_packed(cfg, packed, device, src=packed_src_rank), device))
        string_path = getattr(config, "string_path", "/feat_org_lf0")
        t_total, n_max, n_tot = 0.0, 0, 0
        for ids, batch_h, n_samples_list, utt_index in decode_batches(feat_list, args.batch_size, string_path,
                                                                      config.upsampling_factor, global_index):
            model.noise_utterance_ids = utt_index
            aux = torch.FloatTensor(batch_h).transpose(1, 2).to(device)
            if kind == "laplace":
                seed = torch.zeros(len(ids), model.seg, device=device)
            else:
                seed = torch.full((len(ids), 1), config.n_quantize // 2, dtype=torch.int64, device=device)
            logging.info("decoding start")
            start = time.time()
            samples_list = model.batch_fast_generate(seed, aux, n_samples_list, args.intervals)
            t_total += time.time() - start
            n_max += max(n_samples_list)
            n_tot += max(n_samples_list) * len(n_samples_list)
            for feat_id, samples in zip(ids, samples_list):
                wav = samples if kind == "laplace" else softmax_mod.decode_mu_law(samples, config.n_quantize)
                write_wav_pcm16(os.path.join(args.outdir, feat_id + ".wav"), np.clip(wav, -1, 1), args.fs)
                logging.info("wrote %s.wav in %s." % (feat_id, args.outdir))
        if n_max:
            logging.info("average time / sample = %.6f sec (%ld samples) [%.3f kHz/s]" % (
                t_total / n_max, n_max, n_max / (1000 * t_total)))
            logging.info("average throughput / sample = %.6f sec (%ld samples) [%.3f kHz/s]" % (
                t_total / n_tot, n_tot, n_tot / (1000 * t_total)))
        return n_tot, t_total


def make_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("--feats", required=True, type=str, help="list or directory of aux feat files")
    p.add_argument("--checkpoint", required=True, type=str, help="model file")
    p.add_argument("--config", required=True, type=str, help="configure file")
    p.add_argument("--outdir", required=True, type=str, help="directory to save generated samples")
    p.add_argument("--fs", default=22050, type=int, help="sampling rate")
    p.add_argument("--batch_size", default=1, type=int, help="number of batch size in decoding")
    p.add_argument("--n_gpus", default=1, type=int, help="number of gpus")
    p.add_argument("--spk_trg", default=None, type=str)
    p.add_argument("--min_idx", default=None, type=int)
    p.add_argument("--intervals", default=4410, type=int, help="log interval")
    p.add_argument("--seed", default=1, type=int, help="seed number")
    p.add_argument("--GPU_device", default=0, type=int, help="selection of GPU device")
    p.add_argument("--GPU_device_str", default=None, type=str, help="selection of GPU device")
    p.add_argument("--verbose", default=1, type=int, help="log level")
    p.add_argument("--noise_source", default=None, choices=["host", "device"],
                   help="not a reference flag: where the sampling noise is drawn - host = the torch CPU generator in "
                        "the reference's order (reproduces the reference's CPU decode), device = inside the kernels; "
                        "default: the model's own default (Laplace host, softmax device)")
    p.add_argument("--plan_only", action="store_true",
                   help="not a reference flag: stop after listing, sharding and the parameter broadcast and write "
                        "<outdir>/decode.<rank>.plan.json (needs no GPU: what the CPU tests of the fan-out drive)")
    return p


def _write_plan(kind: str, args, config, rank: int, world: int, shard, index, rng_key: int) -> int:
    """--plan_only: everything a rank does before its first kernel - checkpoint -> packed buffer on rank 0, the one
    broadcast, the batches of its shard with their global utterance indices - written as JSON, no GPU involved."""
    cfg = build_model(kind, config)._cfg
    packed = None
    if rank == 0:
        model = build_model(kind, config)
        model.load_state_dict(artefacts.load_checkpoint(args.checkpoint)["model"])
        packed = pack_state_dict(cfg, model.state_dict())
    buf = D.broadcast_packed(cfg, packed, "cpu")
    string_path = getattr(config, "string_path", "/feat_org_lf0")
    batches = [{"ids": ids, "n_samples": n, "utt_index": ui}
               for ids, _, n, ui in decode_batches(shard, args.batch_size, string_path, config.upsampling_factor, index)]
    with open(os.path.join(args.outdir, f"decode.{rank}.plan.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "shard": list(shard), "index": index, "rng_key": int(rng_key),
                   "packed_numel": int(buf.numel()), "packed_sum": float(buf.double().sum()), "batches": batches}, f)
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


def _rank_main(kind: str, argv, rank: int, world: int, port: int, visible: str) -> None:
    """entry of one spawned rank: the torchrun environment contract, then `main` (never returns: exits with its code)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if visible:
        # the reference exports CUDA_VISIBLE_DEVICES = GPU_device_str and uses GPU r of that list (decode_cswnv...py:150-154,263)
        os.environ["HIP_VISIBLE_DEVICES"] = visible
    sys.exit(main(kind, argv))


def fan_out(kind: str, argv, n_gpus: int, visible: str = None) -> int:
    """`--n_gpus N` without a launcher (run.sh:675-684): one spawned process per shard, rendezvous on 127.0.0.1, wait for
    all (decode_cswnv...py:261-274).  The parent makes no GPU call.  -> 0, or 1 if any rank failed (the others are ended:
    they would wait in the broadcast / barrier for ever)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(kind, list(argv) if argv is not None else sys.argv[1:], r, n_gpus, port,
                                                  visible)) for r in range(n_gpus)]
    for p in procs:
        p.start()
    failed = False
    live = list(procs)
    while live:
        for p in list(live):
            p.join(timeout=0.1)
            if p.exitcode is None:
                continue
            live.remove(p)
            if p.exitcode != 0 and not failed:
                failed = True
                for q in live:
                    q.terminate()
    return 1 if failed else 0


def main(kind: str, argv=None) -> int:
    args = make_parser().parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.n_gpus > 1:
        return fan_out(kind, argv, args.n_gpus, args.GPU_device_str)        # before anything below can touch a GPU
    rank, world, local = D.init_from_env()
    os.makedirs(args.outdir, exist_ok=True)
    level = logging.INFO if args.verbose > 0 else logging.WARN
    logging.basicConfig(level=level, format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s",
                        datefmt="%m/%d/%Y %I:%M:%S", filename=os.path.join(args.outdir, f"decode.{rank}.log"
                                                                       if world > 1 else "decode.log"))
    logging.getLogger().addHandler(logging.StreamHandler())
    os.environ["PYTHONHASHSEED"] = str(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    config = load_config(args.config)
    logging.info(config)
    try:
        feat_list = list_features(args.feats, getattr(config, "string_path", "/feat_org_lf0"))
    except FileNotFoundError:
        logging.error("--feats should be directory or list.")
        return 1
    if world > 1 and args.n_gpus != world:
        logging.warning("--n_gpus %d ignored: running under a launcher with %d ranks", args.n_gpus, world)
    # the one key of the in-kernel noise generator, identical on every rank; from a generator of its own so that the
    # host stream (module init draws, host-drawn noise) stays what the reference's script would draw after manual_seed
    from . import noise as _noise
    rng_key = _noise.draw_rng_seed(torch.Generator().manual_seed(args.seed))
    index = [int(i) for i in D.shard_utterances(range(len(feat_list)), world)[rank]]   # np.array_split, decode_cswnv...py:200-201
    shard = [feat_list[i] for i in index]
    if args.plan_only:
        return _write_plan(kind, args, config, rank, world, shard, index, rng_key)
    if not torch.cuda.is_available():
        logging.error("no HIP device: the MI355X build has no CPU path")
        return 1
    n_dev = torch.cuda.device_count()
    if world == 1 and args.GPU_device_str is not None:
        local = int(args.GPU_device_str.split(",")[0])
    elif world == 1:
        local = args.GPU_device
    elif local >= n_dev:
        local = local % n_dev            # more ranks than visible GPUs (a one-GPU test box): share; the broadcast then runs over gloo
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    gpu_decode(kind, args, config, shard, device, global_index=index, rng_key=rng_key)
    D.barrier(device)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0
