"""ctypes binding of the C ABI in include/swn_hip.h (libswn_hip.so, built in-tree by
csrc/Makefile with hipcc --offload-arch=gfx950).

There is deliberately no fallback: if the library cannot be loaded the product path raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_void_p

from .config import NetConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SWN_HIP_LIB", os.path.join(_HERE, "libswn_hip.so"))   # override: diagnostic builds
CSRC = os.path.join(_HERE, "csrc")

_lock = threading.Lock()
_lib = None


class NetDesc(ctypes.Structure):
    """mirror of `swn_net_desc` (include/swn_hip.h)."""
    _fields_ = [(n, c_int32) for n in (
        "kind", "n_aux", "hid_chn", "skip_chn", "aux_kernel_size", "aux_dilation_size",
        "dilation_depth", "dilation_repeat", "kernel_size", "upsampling_factor", "seg", "lpc",
        "n_quantize", "wav_conv_flag", "audio_in_flag", "aux_conv2d_flag")]


class DecodeIO(ctypes.Structure):
    """mirror of `swn_decode_io` (include/swn_hip.h): inputs of the sampling loop."""
    _fields_ = [("noise_dev", c_void_p), ("forced_dev", c_void_p), ("seed_dev", c_void_p), ("noise_out_dev", c_void_p),
                ("rng_seed", ctypes.c_uint64), ("rng_utt0", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("rng_utt_ids_dev", c_void_p)]


ABI_VERSION = 3
PRECISION_FP32, PRECISION_BF16 = 0, 1          # SWN_PRECISION_* (include/swn_hip.h)


def desc_from_cfg(cfg: NetConfig) -> NetDesc:
    soft = cfg.kind == "softmax"
    return NetDesc(kind=1 if soft else 0, n_aux=cfg.n_aux, hid_chn=cfg.hid_chn, skip_chn=cfg.skip_chn,
                   aux_kernel_size=cfg.aux_kernel_size, aux_dilation_size=cfg.aux_dilation_size,
                   dilation_depth=cfg.dilation_depth, dilation_repeat=cfg.dilation_repeat,
                   kernel_size=cfg.kernel_size, upsampling_factor=cfg.upsampling_factor,
                   seg=1 if soft else cfg.seg, lpc=0 if soft else cfg.lpc,
                   n_quantize=cfg.n_quantize if soft else 0,
                   wav_conv_flag=int(cfg.wav_conv_flag), audio_in_flag=int(cfg.audio_in_flag and soft),
                   aux_conv2d_flag=int(cfg.aux_conv2d_flag and not soft))


# name -> (restype, argtypes); must list every symbol include/swn_hip.h declares
SIGNATURES = {
    "swn_abi_version": (c_int, []),
    "swn_strerror": (c_char_p, [c_int]),
    "swn_last_error_detail": (c_char_p, []),
    "swn_device_count": (c_int, []),
    "swn_receptive_field": (c_int, [POINTER(NetDesc)]),
    "swn_num_tensors": (c_int, [POINTER(NetDesc)]),
    "swn_packed_floats": (c_size_t, [POINTER(NetDesc)]),
    "swn_layout_offsets": (c_int, [POINTER(NetDesc), POINTER(c_size_t), c_int]),
    "swn_pack_params": (c_int, [POINTER(NetDesc), POINTER(c_void_p), c_int, c_void_p, c_size_t]),
    "swn_pack_params_device": (c_int, [POINTER(NetDesc), POINTER(c_void_p), c_int, c_void_p, c_size_t, c_void_p]),
    "swn_frontend_work_floats": (c_size_t, [POINTER(NetDesc), c_int, c_int]),
    "swn_cond_floats": (c_size_t, [POINTER(NetDesc), c_int, c_int]),
    "swn_frontend": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "swn_decode_state_floats": (c_size_t, [POINTER(NetDesc), c_int]),
    "swn_decode": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_int, c_int, c_int, POINTER(DecodeIO),
                           c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "swn_forward_work_floats": (c_size_t, [POINTER(NetDesc), c_int, c_int]),
    "swn_forward": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                            c_void_p, c_void_p, c_void_p]),
    "swn_bf16_weight_bytes": (c_size_t, [POINTER(NetDesc)]),
    "swn_pack_bf16": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_void_p]),
    "swn_forward_bf16_work_bytes": (c_size_t, [POINTER(NetDesc), c_int, c_int]),
    "swn_forward_bf16": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    "swn_backward_work_floats": (c_size_t, [POINTER(NetDesc), c_int, c_int]),
    "swn_backward": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "swn_drop_fused_path": (c_int, [POINTER(NetDesc), c_int, c_int, c_void_p]),
    "swn_forward_drop_work_floats": (c_size_t, [POINTER(NetDesc), c_int, c_int]),
    "swn_forward_drop": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "swn_unfold_grads_device": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "swn_forward_bf16_keep_floats": (c_size_t, [POINTER(NetDesc), c_int, c_int]),
    "swn_forward_bf16_keep": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "swn_backward_keep": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "swn_backward_bf16_work_floats": (c_size_t, [POINTER(NetDesc), c_int, c_int]),
    "swn_backward_bf16": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "swn_backward_drop_work_floats": (c_size_t, [POINTER(NetDesc), c_int, c_int]),
    "swn_backward_drop": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "swn_bf16_train_forward_supported": (c_int, [POINTER(NetDesc)]),
    "swn_bf16_work_to_f32": (c_int, [POINTER(NetDesc), c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "swn_laplace_head_backward": (c_int, [POINTER(NetDesc), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "swn_laplace_head": (c_int, [POINTER(NetDesc), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
}


STAMP_PATH = os.path.join(_HERE, "libswn_hip.so.srchash")


def source_hash() -> str:
    """sha256 over everything the library is built from: csrc/* sources + Makefile + the public header."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp")) or f == "Makefile")
    for path in [os.path.join(CSRC, f) for f in files] + [os.path.join(os.path.dirname(_HERE), "include", "swn_hip.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def is_stale() -> bool:
    """True when the in-tree .so is missing or was not built from the sources now in the tree (the .so and its
    hash stamp are git-ignored but travel to the GPU box, so a stale binary would otherwise go unnoticed)."""
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != source_hash()


def build(force: bool = False) -> str:
    """compile the HIP library in-tree (hipcc cross-compiles gfx950 without a GPU).  A library whose recorded source
    hash differs from the tree is rebuilt from scratch; `force` rebuilds regardless."""
    if force or is_stale():
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libswn_hip.so failed:\n" + r.stdout + r.stderr)
    write_stamp()
    return LIB_PATH


def write_stamp() -> None:
    """record which sources the in-tree .so was built from (also called by csrc/Makefile after linking)."""
    with open(STAMP_PATH, "w") as f:
        f.write(source_hash() + "\n")


def lib() -> ctypes.CDLL:
    """load (building if absent) libswn_hip.so; raises RuntimeError when unavailable."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            build()
        elif is_stale():
            # a binary that was not built from the sources in the tree: rebuild where a compiler exists, else say so
            import shutil
            import warnings
            if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
                build()
            else:
                warnings.warn(f"{LIB_PATH} was not built from the sources in this tree (source hash differs) and no hipcc "
                              "is available to rebuild it", RuntimeWarning)
        # torch ships its own libamdhip64; it must be in the process BEFORE our library so that both
        # resolve to ONE HIP runtime (device pointers and streams come from torch).  Loading ours
        # first binds /opt/rocm's copy, which then reports "no ROCm-capable device".
        import torch  # noqa: F401
        try:
            l = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise RuntimeError(f"shallow_wavenet_amd: cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.swn_abi_version() != ABI_VERSION:
            raise RuntimeError("shallow_wavenet_amd: ABI version mismatch")
        _lib = l
        return l


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().swn_strerror(rc).decode()
        detail = lib().swn_last_error_detail().decode()
        raise RuntimeError(f"swn_hip {what}: {msg} ({rc}) {detail}")
