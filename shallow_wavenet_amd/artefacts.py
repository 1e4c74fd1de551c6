"""Stage artefacts shared by the training and decode drivers: `model.conf` and `checkpoint-N.pkl`.

The reference writes `torch.save(args, expdir + "/model.conf")` (the argparse Namespace,
train_cswnv_laplace-stftcmplx_shift1.py:293) and per-epoch checkpoints
`{"model", "optimizer", "numpy_random_state", "torch_random_state", "iterations"}` (:162-182); stage 5/8
reads `config.<field>` attributes and `torch.load(checkpoint)["model"]`
(decode_cswnv_laplace-shift1.py:208-224, run.sh:658,765,831 hand it `checkpoint-${min_idx}.pkl`).

Everything here loads with `weights_only=True` (nothing in a file is executed): the Namespace and the
numpy reconstruction helpers are allow-listed explicitly, so files written by the reference's own scripts
load as well as ours.  Our own checkpoints store the numpy MT19937 state as plain Python / tensor values
under the reference's key, so they need no allow-list at all.
"""
from __future__ import annotations

import argparse
import json
import os
from types import SimpleNamespace
from typing import Any, Dict

import numpy as np
import torch


def _numpy_safe_globals():
    """what a pickled ndarray / numpy scalar needs (the reference stores np.random.get_state() raw)."""
    out = [np.ndarray, np.dtype]
    try:
        from numpy._core import multiarray as _ma          # numpy >= 2
    except ImportError:                                     # pragma: no cover
        from numpy.core import multiarray as _ma
    out += [_ma._reconstruct, _ma.scalar]
    # files written under numpy 1.x (the reference pins torch 1.3.1 / numpy of that era) name the same helpers by their
    # old module path
    out += [(_ma._reconstruct, "numpy.core.multiarray._reconstruct"), (_ma.scalar, "numpy.core.multiarray.scalar")]
    for name in ("UInt32DType", "Float64DType", "Int64DType", "Float32DType"):
        dt = getattr(getattr(np, "dtypes", None), name, None)
        if dt is not None:
            out.append(dt)
    return out


def _safe_load(path: str):
    with torch.serialization.safe_globals([argparse.Namespace, SimpleNamespace] + _numpy_safe_globals()):
        return torch.load(path, map_location="cpu", weights_only=True)


# ----------------------------------------------------------------------------------------- model.conf
def save_config(args: argparse.Namespace, path: str) -> None:
    """the argparse Namespace itself, like the reference (train_cswnv...py:293)."""
    if not isinstance(args, argparse.Namespace):
        args = argparse.Namespace(**dict(args))
    torch.save(args, path)


def load_config(path: str):
    """model.conf -> object with the constructor fields as attributes.  Accepts the Namespace our
    training drivers and the reference's write, a plain dict (round-1 files), or the fields as JSON."""
    if path.endswith(".json"):
        with open(path) as f:
            return SimpleNamespace(**json.load(f))
    conf = _safe_load(path)
    if isinstance(conf, dict):
        conf = argparse.Namespace(**conf)
    return conf


# ----------------------------------------------------------------------------------------- checkpoints
def pack_numpy_random_state(state) -> tuple:
    """np.random.get_state() -> the same 5-tuple with the key vector as an int64 tensor."""
    name, keys, pos, has_gauss, cached = state
    return (str(name), torch.from_numpy(np.asarray(keys, dtype=np.int64)), int(pos), int(has_gauss), float(cached))


def unpack_numpy_random_state(state) -> tuple:
    """inverse of pack_numpy_random_state; a raw numpy tuple (reference-written file) passes through."""
    name, keys, pos, has_gauss, cached = state
    if isinstance(keys, torch.Tensor):
        keys = keys.numpy()
    return (str(name), np.asarray(keys, dtype=np.uint32), int(pos), int(has_gauss), float(cached))


def save_checkpoint(checkpoint_dir: str, model, optimizer, numpy_random_state, torch_random_state,
                    iterations: int) -> str:
    """the reference's dictionary and file name (train_cswnv...py:162-182); tensors saved on the CPU."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    path = os.path.join(checkpoint_dir, "checkpoint-%d.pkl" % iterations)
    torch.save({"model": sd, "optimizer": optimizer.state_dict(),
                "numpy_random_state": pack_numpy_random_state(numpy_random_state),
                "torch_random_state": torch_random_state, "iterations": iterations}, path)
    return path


def load_checkpoint(path: str) -> Dict[str, Any]:
    """checkpoint-N.pkl / checkpoint-final.pkl -> dict (at least "model"); `numpy_random_state`, when
    present, comes back in the form np.random.set_state takes."""
    ck = _safe_load(path)
    if not isinstance(ck, dict) or "model" not in ck:
        raise RuntimeError(f"{path}: not a checkpoint (no 'model' entry)")
    if ck.get("numpy_random_state") is not None:
        ck["numpy_random_state"] = unpack_numpy_random_state(ck["numpy_random_state"])
    return ck
