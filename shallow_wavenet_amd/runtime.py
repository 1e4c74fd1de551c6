"""Host runtime over the C ABI: packed parameters in HBM, scratch buffers, launches.

torch is used for device memory, streams and (in `dist.py`) RCCL only; every arithmetic
step of the hot path is a HIP kernel behind `include/swn_hip.h`.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import ops as _ops          # registers torch.ops.swn.*
from .config import NetConfig

_O = torch.ops.swn


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def pack_state_dict(cfg: NetConfig, state_dict: Dict[str, "np.ndarray | torch.Tensor"]) -> torch.Tensor:
    """reference state_dict (any device) -> flat fp32 host tensor in kernel layout
    (swn_pack_params; layout documented in csrc/swn_geom.hpp)."""
    L = _lib.lib()
    desc = _lib.desc_from_cfg(cfg)
    shapes = cfg.param_shapes()
    n = L.swn_num_tensors(ctypes.byref(desc))
    if n < 0:
        _lib.check(n, "descriptor")
    if n != len(shapes):
        raise RuntimeError("tensor count mismatch between host config and library")
    host = []
    for name, shp in shapes:
        if name not in state_dict:
            raise KeyError(f"state_dict is missing {name}")
        v = state_dict[name]
        v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        if tuple(v.shape) != tuple(shp):
            raise RuntimeError(f"{name}: shape {tuple(v.shape)} != reference shape {tuple(shp)}")
        host.append(np.ascontiguousarray(v, dtype=np.float32))
    ptrs = (ctypes.c_void_p * n)(*[h.ctypes.data for h in host])
    total = L.swn_packed_floats(ctypes.byref(desc))
    packed = torch.zeros(total, dtype=torch.float32)
    _lib.check(L.swn_pack_params(ctypes.byref(desc), ptrs, n, ctypes.c_void_p(packed.data_ptr()), total),
               "pack_params")
    return packed


def pack_parameters_device(cfg: NetConfig, tensors: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None
                           ) -> torch.Tensor:
    """reference-layout parameter tensors ALREADY ON THE DEVICE (state_dict order) -> packed buffer on that device
    (swn_pack_params_device): one launch, no host round trip; `out` is reused when given."""
    L = _lib.lib()
    desc = _lib.desc_from_cfg(cfg)
    shapes = cfg.param_shapes()
    if len(tensors) != len(shapes):
        raise RuntimeError(f"{len(tensors)} tensors given, the configuration has {len(shapes)}")
    dev = tensors[0].device
    keep = []
    for t, (name, shp) in zip(tensors, shapes):
        if tuple(t.shape) != tuple(shp):
            raise RuntimeError(f"{name}: shape {tuple(t.shape)} != reference shape {tuple(shp)}")
        if t.device != dev or dev.type != "cuda":
            raise RuntimeError(f"{name}: device-side packing needs every parameter on one HIP device")
        t = t.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        keep.append(t)
    total = L.swn_packed_floats(ctypes.byref(desc))
    if out is None:
        out = torch.empty(total, dtype=torch.float32, device=dev)
    elif out.numel() != total or out.device != dev or out.dtype != torch.float32:
        raise RuntimeError("packed output buffer has the wrong size / device")
    ptrs = (ctypes.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
    with torch.cuda.device(dev):
        _lib.check(L.swn_pack_params_device(ctypes.byref(desc), ptrs, len(keep), _ptr(out), total, _stream_ptr(dev)),
                   "pack_params_device")
    return out


LAYOUT_FIELDS = ("scale_w", "scale_b", "aux_w0", "aux_w1", "aux_w2", "aux_w3", "aux_b0", "aux_b1", "aux_b2", "aux_b3",
                 "wx", "wxa", "wup", "bup", "bx", "cb", "cv", "cc", "ct", "wd", "bd", "wsk", "bsk", "w1", "b1", "w2",
                 "b2", "total", "bxr")


_precision = _lib.PRECISION_FP32      # host-side default handed to the training entry points (the C ABI holds no mode)


def current_precision() -> int:
    """SWN_PRECISION_* value a training-mode forward started now would be run in."""
    return _precision


class train_precision:
    """context manager / setter for the arithmetic of the training contractions: "fp32" = exact fp32 MFMA (parity mode,
    default), "bf16" = bf16 operands with fp32 accumulation.  Python-side sugar only: the C entry points take the mode
    as an argument (include/swn_hip.h SWN_PRECISION_*); a forward records the mode it ran in and its backward is
    issued in that same mode whatever is current by then."""
    MODES = {"fp32": _lib.PRECISION_FP32, "bf16": _lib.PRECISION_BF16}

    def __init__(self, mode: str):
        global _precision
        if mode not in self.MODES:
            raise ValueError(f"train precision must be one of {sorted(self.MODES)}")
        self._prev = _precision
        _precision = self.MODES[mode]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        global _precision
        _precision = self._prev
        return False


def layout_offsets(cfg: NetConfig) -> Dict[str, int]:
    """float offsets of the packed sections (swn_layout_offsets)."""
    d = _lib.desc_from_cfg(cfg)
    buf = (ctypes.c_size_t * len(LAYOUT_FIELDS))()
    n = _lib.lib().swn_layout_offsets(ctypes.byref(d), buf, len(LAYOUT_FIELDS))
    if n != len(LAYOUT_FIELDS):
        _lib.check(n if n < 0 else -2, "layout_offsets")
    return {k: int(buf[i]) for i, k in enumerate(LAYOUT_FIELDS)}


class HipNet:
    """One network's packed parameters resident in HBM plus the launch helpers."""

    def __init__(self, cfg: NetConfig, packed: torch.Tensor, device):
        if not torch.cuda.is_available():
            raise RuntimeError("shallow_wavenet_amd needs a HIP device (no CPU fallback)")
        self.cfg = cfg
        self.desc = _lib.desc_from_cfg(cfg)
        self.dlist = _ops.desc_list(cfg)          # the same descriptor as the integer list the custom ops take
        self.device = torch.device(device)
        self.lib = _lib.lib()
        self.packed = packed.to(self.device, non_blocking=False).contiguous()
        self.packed_version = 0          # bumped by repack(): derived copies (bf16 weights) follow it
        self.fused_backward = True       # mixed-precision mode, BL6 class: swn_backward_bf16 (False: the generic chain)
        self.keep_preactivations = True  # mixed-precision mode, GEMM-stack class: swn_forward_bf16_keep + swn_backward_keep

    def repack(self, tensors: Sequence[torch.Tensor]) -> None:
        """refresh the packed buffer IN PLACE from the live parameter tensors on the device (after an optimizer
        step); launches are stream-ordered behind whatever still reads the old values.  The per-tensor validation of
        pack_parameters_device is done once per set of storages: an optimizer step changes values, not addresses."""
        key = tuple(t.data_ptr() for t in tensors)
        cached = getattr(self, "_repack_cache", None)
        if cached is None or cached[0] != key or any(t.dtype != torch.float32 or not t.is_contiguous() for t in tensors):
            pack_parameters_device(self.cfg, tensors, out=self.packed)       # validates shapes / devices / dtypes
            if all(t.dtype == torch.float32 and t.is_contiguous() for t in tensors):
                self._repack_cache = (key, (ctypes.c_void_p * len(key))(*key))
        else:
            with _ops._on(self.device):
                _lib.check(self.lib.swn_pack_params_device(ctypes.byref(self.desc), cached[1], len(key), _ptr(self.packed),
                                                           self.packed.numel(), _stream_ptr(self.device)), "pack_params_device")
        self.packed_version += 1

    @classmethod
    def from_state_dict(cls, cfg: NetConfig, state_dict, device) -> "HipNet":
        return cls(cfg, pack_state_dict(cfg, state_dict), device)

    # ------------------------------------------------------------------ front end
    def frontend(self, aux: torch.Tensor) -> torch.Tensor:
        """aux (B, n_aux, Tf) fp32 on device -> cond (B, Tf, L*seg*2H)   (torch.ops.swn.frontend)."""
        cond, work = _O.frontend(self.packed, aux.to(self.device), self.dlist)
        self._last_frontend_work = work            # kept for swn_backward (conv_aux activations)
        return cond

    # ------------------------------------------------------------------ decode
    def decode(self, aux: torch.Tensor, n_steps: int, noise: Optional[torch.Tensor] = None,
               forced: Optional[torch.Tensor] = None, want_heads: bool = False, variant: int = 0,
               cond: Optional[torch.Tensor] = None, seed: Optional[torch.Tensor] = None, rng_seed: int = 0,
               rng_utt0: int = 0, want_noise: bool = False, utt_ids: Optional[Sequence[int]] = None):
        """run prologue + n_steps generation steps for every utterance of the batch.

        noise: laplace (B, n_steps, seg) | softmax (B, n_steps, Q), fp32, utterance-major - the host-drawn stream of
               the parity mode; None = the kernels draw it themselves (counter-based generator keyed by `rng_seed`,
               utterance b drawing as global utterance `utt_ids[b]` when given, else `rng_utt0 + b`).
        seed:  the seed waveform `audio` of batch_fast_generate: laplace (B, seg) fp32 | softmax (B,) classes; None =
               zeros / class Q/2.
        returns (out, heads): out laplace (B, n_steps*seg) fp32 | softmax (B, n_steps) int32; with want_noise=True
        a third value: the noise the kernels used, in the layout of `noise`.
        """
        if cond is None:
            cond = self.frontend(aux)
        ids = None if utt_ids is None else torch.as_tensor(list(utt_ids), dtype=torch.int32)
        out, heads, used = _O.decode(self.packed, cond, noise, forced, seed, self.dlist, int(n_steps), int(variant),
                                     int(rng_seed) & 0x7FFFFFFFFFFFFFFF, int(rng_utt0) & 0xFFFFFFFF, bool(want_heads),
                                     bool(want_noise), ids)
        heads = heads if want_heads else None
        if want_noise:
            return out, heads, used
        return out, heads

    # ------------------------------------------------------------------ teacher-forced stack
    def forward(self, aux: torch.Tensor, audio: torch.Tensor, want_hidden: bool = False,
                cond: Optional[torch.Tensor] = None):
        """raw out_2 outputs (B, n_out, Tp) of the teacher-forced stack (torch.ops.swn.stack_forward).
        audio: laplace (B, 1, T-seg) fp32 | softmax (B, T-1) integer indices."""
        if cond is None:
            cond = self.frontend(aux)
        out, _work, hs = _O.stack_forward(self.packed, cond, audio.to(self.device), self.dlist, bool(want_hidden))
        return out, (hs if want_hidden else None)

    def forward_bf16(self, aux: torch.Tensor, audio: torch.Tensor, cond: Optional[torch.Tensor] = None):
        """bf16 MFMA variant of `forward` (BL6-class Laplace nets and H%64==0 nets of either kind): raw (B, n_out, Tp) fp32."""
        if cond is None:
            cond = self.frontend(aux)
        if getattr(self, "_wbf16", None) is None or getattr(self, "_wbf16_version", -1) != self.packed_version:
            self._wbf16 = _O.pack_bf16(self.packed, self.dlist)
            self._wbf16_version = self.packed_version
        out, _work = _O.stack_forward_bf16(self.packed, self._wbf16, cond, audio.to(self.device), self.dlist)
        return out

    def laplace_head(self, raw: torch.Tensor, clip: bool = False):
        """raw (B, n_out, Tp) -> (mu, b, logb, a, b_clip, logb_clip, below_floor) time-major (torch.ops.swn.laplace_head)."""
        mu, b, logb, a, bc, lc, flag = _O.laplace_head(raw, self.dlist, bool(clip))
        return mu, b, logb, (a if self.cfg.lpc > 0 else None), (bc if clip else None), (lc if clip else None), flag


    # ------------------------------------------------------------------ training (fp32)
    def forward_train(self, aux: torch.Tensor, audio: torch.Tensor, drop=None):
        """forward that keeps what swn_backward needs: returns (raw, saved).  `drop` = (drop_x, [drop_h per layer or
        None]) switches to the dropout-mode kernels (noise.dropout_masks draws them in the reference's order)."""
        if drop is not None:
            return self._forward_train_drop(aux, audio, drop)
        cfg = self.cfg
        aux = aux.to(self.device, torch.float32).contiguous()
        cond, fe_work = _ops.frontend_impl(self.packed, aux, self.dlist)
        soft = cfg.kind == "softmax"
        B, Tf = cond.shape[0], cond.shape[1]
        T = Tf * cfg.U
        Tp = T - 1 if soft else T - 2 * cfg.seg + 1
        audio = audio.to(self.device, torch.int32 if soft else torch.float32).contiguous()
        mode = current_precision()
        if mode == _lib.PRECISION_BF16:
            res = self._bf16_train_forward(cond, audio, B, Tf)
            if res is not None:
                out, wb, work = res
                return out, dict(aux=aux, cond=cond, fe_work=fe_work, audio=audio, work=work, work_bf16=wb, a_keep=self._a_keep,
                                 B=B, Tf=Tf, precision=mode, packed_version=self.packed_version)
        out, work, _ = _O.stack_forward(self.packed, cond, audio, self.dlist, False)
        return out, dict(aux=aux, cond=cond, fe_work=fe_work, audio=audio, work=work, B=B, Tf=Tf, precision=mode,
                         packed_version=self.packed_version)

    def _bf16_train_forward(self, cond, audio, B, Tf):
        """mixed-precision mode: bf16 forward -> (raw, bf16 work buffer, fp32 work buffer or None).  Returns None (caller
        runs the fp32 forward) where the library has no bf16 stack for the geometry.  The fp32 expansion of the bf16
        activations (what swn_backward reads) is skipped where the backward will read the bf16 buffer itself
        (swn_backward_bf16, BL6 class); `backward` makes it on demand if that choice is revoked in between.
        The bf16 copy of the weights follows `packed_version` (bumped whenever the parameters were re-packed)."""
        L = self.lib
        d = ctypes.byref(self.desc)
        if L.swn_bf16_train_forward_supported(d) != 1:
            return None
        nbytes = L.swn_bf16_weight_bytes(d)
        wb = torch.empty(L.swn_forward_bf16_work_bytes(d, B, Tf), dtype=torch.uint8, device=self.device)
        if getattr(self, "_wbf16", None) is None:
            self._wbf16 = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        soft = self.cfg.kind == "softmax"
        Tp = Tf * self.cfg.U - 1 if soft else Tf * self.cfg.U - 2 * self.cfg.seg + 1
        out = torch.empty((B, self.cfg.n_out, Tp), dtype=torch.float32, device=self.device)
        with _ops._on(self.device):
            st = _stream_ptr(self.device)
            if getattr(self, "_wbf16_version", -1) != self.packed_version:
                _lib.check(L.swn_pack_bf16(d, _ptr(self.packed), _ptr(self._wbf16), st), "pack_bf16")
                self._wbf16_version = self.packed_version
            nkeep = L.swn_forward_bf16_keep_floats(d, B, Tf) if self.keep_preactivations else 0
            if nkeep:
                # GEMM-stack geometries: the forward also keeps the gate pre-activations, the backward skips its recompute GEMMs
                self._a_keep = torch.empty(nkeep, dtype=torch.float32, device=self.device)
                _lib.check(L.swn_forward_bf16_keep(d, _ptr(self.packed), _ptr(self._wbf16), _ptr(cond), _ptr(audio), B, Tf,
                                                   _ptr(wb), _ptr(out), _ptr(self._a_keep), st), "forward_bf16_keep")
            else:
                self._a_keep = None
                _lib.check(L.swn_forward_bf16(d, _ptr(self.packed), _ptr(self._wbf16), _ptr(cond), _ptr(audio), B, Tf,
                                              _ptr(wb), _ptr(out), st), "forward_bf16")
        fused = self.fused_backward and _ops.backward_bf16_supported(self.dlist, B, Tf)
        return out, wb, (None if fused else self._expand_bf16_work(wb, B, Tf, _lib.PRECISION_BF16))

    def _expand_bf16_work(self, wb, B, Tf, mode):
        """bf16 work buffer of swn_forward_bf16 -> the fp32 work layout swn_backward reads (swn_bf16_work_to_f32)."""
        L, d = self.lib, ctypes.byref(self.desc)
        work = torch.empty(L.swn_forward_work_floats(d, B, Tf), dtype=torch.float32, device=self.device)
        with _ops._on(self.device):
            _lib.check(L.swn_bf16_work_to_f32(d, _ptr(self.packed), _ptr(wb), B, Tf, _ptr(work), int(mode), _stream_ptr(self.device)),
                       "bf16_work_to_f32")
        return work

    def _drop_args(self, drop):
        drop_x, drop_h = drop
        cfg = self.cfg
        drop_x = drop_x.to(self.device, torch.float32).contiguous()
        drop_h = [None if m is None else m.to(self.device, torch.float32).contiguous() for m in drop_h]
        if len(drop_h) != cfg.L:
            raise ValueError("drop_h needs one entry (mask or None) per layer")
        ptrs = (ctypes.c_void_p * cfg.L)(*[_ptr(m) for m in drop_h])
        return drop_x, drop_h, ptrs

    def _forward_train_drop(self, aux, audio, drop):
        cfg = self.cfg
        aux = aux.to(self.device, torch.float32).contiguous()
        soft = cfg.kind == "softmax"
        B, Tf = aux.shape[0], aux.shape[2]
        d = ctypes.byref(self.desc)
        if aux.shape[1] != cfg.n_aux:
            raise RuntimeError(f"aux has {aux.shape[1]} channels, model expects {cfg.n_aux}")
        # conv_aux activations only: aux_drop acts at sample rate, the hoisted in_x products (cond) are not used
        fe_work = torch.empty(self.lib.swn_frontend_work_floats(d, B, Tf), dtype=torch.float32, device=self.device)
        with _ops._on(self.device):
            _lib.check(self.lib.swn_frontend(d, _ptr(self.packed), _ptr(aux), B, Tf, _ptr(fe_work), _ptr(None),
                                             _stream_ptr(self.device)), "frontend")
        T = Tf * cfg.U
        Tp = T - 1 if soft else T - 2 * cfg.seg + 1
        coff = 1 if soft else cfg.seg
        audio = audio.to(self.device, torch.int32 if soft else torch.float32).contiguous()
        drop_x, drop_h, ptrs = self._drop_args(drop)
        if tuple(drop_x.shape) != (B, cfg.A0, T - coff) or any(m is not None and tuple(m.shape) != (B, cfg.H, Tp) for m in drop_h):
            raise ValueError("dropout mask shapes must be (B, A0, T-coff) and (B, H, Tp)")
        work = torch.empty(self.lib.swn_forward_drop_work_floats(d, B, Tf), dtype=torch.float32, device=self.device)
        out = torch.empty((B, cfg.n_out, Tp), dtype=torch.float32, device=self.device)
        mode = current_precision()
        with _ops._on(self.device):
            _lib.check(self.lib.swn_forward_drop(d, _ptr(self.packed), _ptr(fe_work), _ptr(audio), B, Tf, _ptr(drop_x),
                                                 ctypes.cast(ptrs, ctypes.c_void_p), _ptr(work), _ptr(out), _ptr(None),
                                                 mode, _stream_ptr(self.device)), "forward_drop")
        return out, dict(aux=aux, cond=None, fe_work=fe_work, audio=audio, work=work, B=B, Tf=Tf,
                         drop=(drop_x, drop_h, ptrs), precision=mode, packed_version=self.packed_version)

    def backward(self, saved, grad_raw: torch.Tensor) -> torch.Tensor:
        """gradient of the loss wrt the packed parameter buffer, given d loss / d raw (B, n_out, Tp).  Runs in the
        arithmetic mode the forward that made `saved` ran in (the work-buffer layout of the dropout mode depends on it).
        Raises if the parameters were re-packed since that forward (an optimizer step or load_state_dict between a
        forward and its backward: torch's own in-place-modification error in the reference)."""
        L, d = self.lib, ctypes.byref(self.desc)
        B, Tf = saved["B"], saved["Tf"]
        mode = int(saved.get("precision", current_precision()))
        if saved.get("packed_version", self.packed_version) != self.packed_version:
            raise RuntimeError("one of the parameters needed for gradient computation has been modified since the forward "
                               "pass (the packed parameter buffer was refreshed in place: version "
                               f"{self.packed_version}, the forward saw {saved['packed_version']})")
        grad_raw = grad_raw.to(self.device, torch.float32).contiguous()
        if saved.get("drop") is not None:
            drop_x, drop_h, ptrs = saved["drop"]
            work = torch.empty(L.swn_backward_drop_work_floats(d, B, Tf), dtype=torch.float32, device=self.device)
            gp = torch.empty_like(self.packed)
            with _ops._on(self.device):
                _lib.check(L.swn_backward_drop(d, _ptr(self.packed), _ptr(saved["aux"]), _ptr(saved["fe_work"]),
                                               _ptr(saved["audio"]), _ptr(saved["work"]), _ptr(None), _ptr(drop_x),
                                               ctypes.cast(ptrs, ctypes.c_void_p), _ptr(grad_raw), B, Tf, _ptr(work),
                                               _ptr(gp), mode, _stream_ptr(self.device)), "backward_drop")
            return gp
        wb = saved.get("work_bf16")
        if (wb is not None and self.fused_backward and mode == _lib.PRECISION_BF16
                and _ops.backward_bf16_supported(self.dlist, B, Tf)):
            # BL6 class after a bf16 forward: the gated layers' backward fused per layer (csrc/swn_bwd_bl6.hip)
            return _ops.stack_backward_bf16_impl(self.packed, saved["aux"], saved["cond"], saved["fe_work"], saved["audio"],
                                                 wb, grad_raw, self.dlist)
        if saved["work"] is None:                      # the forward counted on the fused backward
            saved["work"] = self._expand_bf16_work(wb, B, Tf, mode)
        if saved.get("a_keep") is not None and mode == _lib.PRECISION_BF16:
            work = torch.empty(L.swn_backward_work_floats(d, B, Tf), dtype=torch.float32, device=self.device)
            gp = torch.empty_like(self.packed)
            with _ops._on(self.device):
                _lib.check(L.swn_backward_keep(d, _ptr(self.packed), _ptr(saved["aux"]), _ptr(saved["cond"]), _ptr(saved["fe_work"]),
                                               _ptr(saved["audio"]), _ptr(saved["work"]), _ptr(saved["a_keep"]), _ptr(grad_raw), B, Tf,
                                               _ptr(work), _ptr(gp), _stream_ptr(self.device)), "backward_keep")
            return gp
        return _ops.stack_backward_impl(self.packed, saved["aux"], saved["cond"], saved["fe_work"], saved["audio"],
                                        saved["work"], grad_raw, self.dlist, mode)

    def laplace_head_backward(self, raw, gmu, gb, glogb, ga, gb_clip=None, glogb_clip=None) -> torch.Tensor:
        return _ops.laplace_head_backward_impl(raw, gmu, gb, glogb, ga, gb_clip, glogb_clip, self.dlist)
