"""Multi-GPU decode: one process per GPU, utterance-level sharding, ONE collective.

The reference fans out with `np.array_split(feat_list, n_gpus)` + one forked process per GPU,
each re-reading the checkpoint from disk, and never communicates
(decode_cswnv_laplace-shift1.py:200-201,261-274).  Here rank 0 packs the parameters once and the
flat fp32 buffer (3-38 MB) is broadcast over RCCL/xGMI; after that ranks are independent (no
cross-GPU dependency inside or between utterances), so there is no per-step traffic at all.
On CPU test rigs the same code runs over gloo.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .config import NetConfig


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the process
    group when WORLD_SIZE > 1 (backend nccl == RCCL on ROCm when a GPU is present, else gloo)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # RCCL needs one GPU per rank; more ranks than visible GPUs (rehearsals on a one-GPU box) or no GPU: gloo
            backend = "nccl" if torch.cuda.is_available() and torch.cuda.device_count() >= world else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_utterances(items: Sequence, n_shards: int) -> List[list]:
    """contiguous near-equal split, identical to np.array_split on the unsorted list
    (decode_cswnv_laplace-shift1.py:200-201), so per-GPU batches equal the reference's."""
    return [list(a) for a in np.array_split(np.asarray(list(items), dtype=object), n_shards)]


def packed_size(cfg: NetConfig) -> int:
    d = _lib.desc_from_cfg(cfg)
    return int(_lib.lib().swn_packed_floats(ctypes.byref(d)))


def broadcast_packed(cfg: NetConfig, packed: Optional[torch.Tensor], device, src: int = 0) -> torch.Tensor:
    """every rank returns the packed parameter buffer on `device`; only `src` needs to pass one.
    A single broadcast of one flat tensor (the whole model) - sized for xGMI: one message,
    not one per parameter tensor."""
    n = packed_size(cfg)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        if packed is None:
            raise RuntimeError("single process: the packed buffer must be supplied")
        return packed.to(device)
    via = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device(device)   # gloo moves host memory
    if dist.get_rank() == src:
        if packed is None or packed.numel() != n:
            raise RuntimeError("source rank must supply the packed buffer")
        buf = packed.to(via).contiguous()
    else:
        buf = torch.empty(n, dtype=torch.float32, device=via)
    dist.broadcast(buf, src=src)
    return buf.to(device)


def _reduce_device(device):
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier(device=None) -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        if device is not None and torch.device(device).type == "cuda" and dist.get_backend() != "gloo":
            dist.barrier(device_ids=[torch.device(device).index])
        else:
            dist.barrier()
