"""PyTorch custom ops over the C ABI: `torch.ops.swn.*` (one `torch.library` namespace).

(Each op NAME is registered from a plain function NAME_impl; the training autograd Functions of nets/_autograd.py call
the _impl functions directly - they already sit inside an autograd node, and a Python custom op costs 30-50 us of
dispatch per call, which the BL6 training step notices.)

Every op is a thin, stateless wrapper of one entry point of `include/swn_hip.h`: it takes contiguous device tensors
plus the network descriptor as a list of 16 integers (the fields of `swn_net_desc`, i.e. the reference constructor
arguments), allocates its outputs with torch, and launches on torch's current HIP stream.  Failures of the library
surface as `RuntimeError` (the reference's own convention for bad shapes is whatever torch raises).  Fake (meta)
implementations are registered so the ops trace under `torch.compile` / `FakeTensorMode`; gradients are wired by
`nets/_autograd.py` (the backward entry points are ops of this namespace too).

    torch.ops.swn.pack_params(tensors, desc)                       -> packed
    torch.ops.swn.frontend(packed, aux, desc)                      -> (cond, work)
    torch.ops.swn.decode(packed, cond, noise?, forced?, seed?, desc, n_steps, variant, rng_seed, rng_utt0,
                         want_heads, want_noise, utt_ids?)         -> (out, heads, noise_used)
    torch.ops.swn.stack_forward(packed, cond, audio, desc, want_hidden) -> (raw, work, hidden)
    torch.ops.swn.stack_forward_bf16(packed, wbf16, cond, audio, desc)  -> (raw, work)
    torch.ops.swn.pack_bf16(packed, desc)                          -> wbf16
    torch.ops.swn.laplace_head(raw, desc, clip)                    -> (mu, b, logb, a, b_clip, logb_clip, below_floor)
    torch.ops.swn.stack_backward(packed, aux, cond, fe_work, audio, fwd_work, grad_raw, desc, precision) -> grad_packed
    torch.ops.swn.laplace_head_backward(raw, gmu?, gb?, glogb?, ga?, gb_clip?, glogb_clip?, desc) -> grad_raw
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch
from torch.library import custom_op

from . import _lib
from .config import NetConfig

DESC_FIELDS = [n for n, _ in _lib.NetDesc._fields_]


def desc_list(cfg: NetConfig) -> List[int]:
    d = _lib.desc_from_cfg(cfg)
    return [int(getattr(d, f)) for f in DESC_FIELDS]


_DESC_CACHE: dict = {}


def _desc(vals: Sequence[int]) -> "_lib.NetDesc":
    key = tuple(vals)
    d = _DESC_CACHE.get(key)
    if d is None:
        if len(vals) != len(DESC_FIELDS):
            raise RuntimeError(f"descriptor needs {len(DESC_FIELDS)} integers ({', '.join(DESC_FIELDS)})")
        d = _DESC_CACHE[key] = _lib.NetDesc(**{f: int(v) for f, v in zip(DESC_FIELDS, vals)})
    return d


class _on:
    """`with _on(dev):` = torch.cuda.device(dev) only when dev is not the current device (the context manager costs
    more host time than a launch)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _geom(d) -> Tuple[int, int, int, int, int, int]:
    """(soft, seg, L, H, n_out, out1) from a descriptor."""
    soft = d.kind == 1
    seg = 1 if soft else d.seg
    L = d.dilation_depth * d.dilation_repeat
    n_out = d.n_quantize if soft else 2 * d.seg + d.lpc
    return soft, seg, L, d.hid_chn, n_out, (d.n_quantize if soft else d.skip_chn)


def _need_cuda(t: torch.Tensor, what: str) -> None:
    if t.device.type != "cuda":
        raise RuntimeError(f"swn ops need {what} on a HIP device (no CPU path)")


# ------------------------------------------------------------------------------------------ pack
def pack_params_impl(tensors: List[torch.Tensor], desc: List[int]) -> torch.Tensor:
    """state_dict tensors (reference order, on the device) -> packed buffer (swn_pack_params_device)."""
    L = _lib.lib()
    d = _desc(desc)
    dev = tensors[0].device
    _need_cuda(tensors[0], "the parameters")
    keep = [t.detach().to(torch.float32).contiguous() for t in tensors]
    total = L.swn_packed_floats(ctypes.byref(d))
    out = torch.empty(total, dtype=torch.float32, device=dev)
    ptrs = (ctypes.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
    with _on(dev):
        _lib.check(L.swn_pack_params_device(ctypes.byref(d), ptrs, len(keep), _ptr(out), total, _stream(dev)),
                   "pack_params_device")
    return out


pack_params = custom_op("swn::pack_params", mutates_args=())(pack_params_impl)


@pack_params.register_fake
def _(tensors, desc):
    d = _desc(desc)
    return tensors[0].new_empty(_lib.lib().swn_packed_floats(ctypes.byref(d)), dtype=torch.float32)


# ------------------------------------------------------------------------------------------ front end
def frontend_impl(packed: torch.Tensor, aux: torch.Tensor, desc: List[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """aux (B, n_aux, Tf) -> cond (B, Tf, L*seg*2H) and the work buffer the backward needs (swn_frontend)."""
    L = _lib.lib()
    d = _desc(desc)
    _need_cuda(packed, "the packed parameters")
    dev = packed.device
    aux = aux.to(dev, torch.float32).contiguous()
    B, na, Tf = aux.shape
    if na != d.n_aux:
        raise RuntimeError(f"aux has {na} channels, model expects {d.n_aux}")
    r = ctypes.byref(d)
    work = torch.empty(L.swn_frontend_work_floats(r, B, Tf), dtype=torch.float32, device=dev)
    cond = torch.empty(L.swn_cond_floats(r, B, Tf), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.swn_frontend(r, _ptr(packed), _ptr(aux), B, Tf, _ptr(work), _ptr(cond), _stream(dev)), "frontend")
    return cond.view(B, Tf, -1), work


frontend = custom_op("swn::frontend", mutates_args=())(frontend_impl)


@frontend.register_fake
def _(packed, aux, desc):
    L, d = _lib.lib(), _desc(desc)
    B, _, Tf = aux.shape
    r = ctypes.byref(d)
    n = L.swn_cond_floats(r, B, Tf)
    return packed.new_empty((B, Tf, n // (B * Tf))), packed.new_empty(L.swn_frontend_work_floats(r, B, Tf))


# ------------------------------------------------------------------------------------------ decode
def decode_impl(packed: torch.Tensor, cond: torch.Tensor, noise: Optional[torch.Tensor], forced: Optional[torch.Tensor],
           seed: Optional[torch.Tensor], desc: List[int], n_steps: int, variant: int, rng_seed: int, rng_utt0: int,
           want_heads: bool, want_noise: bool, utt_ids: Optional[torch.Tensor] = None
           ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """prologue + n_steps generation steps for every utterance (swn_decode).  noise None = drawn in the kernels, utterance
    b as global utterance utt_ids[b] (int32 (B), values < 2^31) or rng_utt0 + b."""
    L = _lib.lib()
    d = _desc(desc)
    _need_cuda(packed, "the packed parameters")
    dev = packed.device
    soft, seg, _, _, n_out, _ = _geom(d)
    B, Tf = cond.shape[0], cond.shape[1]
    width = d.n_quantize if soft else seg
    cond = cond.contiguous()
    if noise is not None:
        noise = noise.to(dev, torch.float32).contiguous()
        if tuple(noise.shape) != (B, n_steps, width):
            raise RuntimeError(f"noise shape {tuple(noise.shape)} != {(B, n_steps, width)}")
    if forced is not None:
        forced = forced.to(dev, torch.int32 if soft else torch.float32).contiguous()
        if forced.numel() != B * n_steps * seg:
            raise RuntimeError("forced history has the wrong size")
    if seed is not None:
        seed = seed.to(dev, torch.int32 if soft else torch.float32).contiguous()
        if seed.numel() != B * seg:
            raise RuntimeError(f"seed waveform has {seed.numel()} elements, expected {B * seg}")
    if utt_ids is not None:
        utt_ids = utt_ids.to(dev, torch.int32).contiguous()
        if utt_ids.numel() != B:
            raise RuntimeError(f"utt_ids has {utt_ids.numel()} elements, expected {B}")
    r = ctypes.byref(d)
    state = torch.empty(L.swn_decode_state_floats(r, B), dtype=torch.float32, device=dev)
    out = torch.empty((B, n_steps * seg), dtype=torch.int32 if soft else torch.float32, device=dev)
    heads = torch.empty((B, n_steps, n_out) if want_heads else (0,), dtype=torch.float32, device=dev)
    used = torch.empty((B, n_steps, width) if want_noise else (0,), dtype=torch.float32, device=dev)
    io = _lib.DecodeIO(noise_dev=_ptr(noise), forced_dev=_ptr(forced), seed_dev=_ptr(seed),
                       noise_out_dev=_ptr(used if want_noise else None),
                       rng_seed=int(rng_seed) & 0xFFFFFFFFFFFFFFFF, rng_utt0=int(rng_utt0) & 0xFFFFFFFF, reserved=0,
                       rng_utt_ids_dev=_ptr(utt_ids))
    with _on(dev):
        _lib.check(L.swn_decode(r, _ptr(packed), _ptr(cond), B, Tf, n_steps, ctypes.byref(io), _ptr(state), _ptr(out),
                                _ptr(heads if want_heads else None), variant, _stream(dev)), "decode")
    return out, heads, used


decode = custom_op("swn::decode", mutates_args=())(decode_impl)


@decode.register_fake
def _(packed, cond, noise, forced, seed, desc, n_steps, variant, rng_seed, rng_utt0, want_heads, want_noise, utt_ids=None):
    d = _desc(desc)
    soft, seg, _, _, n_out, _ = _geom(d)
    B = cond.shape[0]
    width = d.n_quantize if soft else seg
    return (packed.new_empty((B, n_steps * seg), dtype=torch.int32 if soft else torch.float32),
            packed.new_empty((B, n_steps, n_out) if want_heads else (0,)),
            packed.new_empty((B, n_steps, width) if want_noise else (0,)))


# ------------------------------------------------------------------------------------------ teacher-forced stack
def _tp(d, Tf: int) -> Tuple[int, int]:
    soft, seg, *_ = _geom(d)
    T = Tf * d.upsampling_factor
    return T, (T - 1 if soft else T - 2 * seg + 1)


def stack_forward_impl(packed: torch.Tensor, cond: torch.Tensor, audio: torch.Tensor, desc: List[int],
                  want_hidden: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """raw out_2 outputs (B, n_out, Tp) of the fp32 parity kernels, the work buffer swn_backward reads, and
    optionally the hidden states (B, L+1, H, Tp) (swn_forward)."""
    Lb = _lib.lib()
    d = _desc(desc)
    _need_cuda(packed, "the packed parameters")
    dev = packed.device
    soft, seg, L, H, n_out, _ = _geom(d)
    B, Tf = cond.shape[0], cond.shape[1]
    T, Tp = _tp(d, Tf)
    audio = audio.to(dev, torch.int32 if soft else torch.float32).contiguous()
    if audio.numel() != B * (T - seg):
        raise RuntimeError(f"audio has {audio.numel()} elements, expected {B * (T - seg)}")
    r = ctypes.byref(d)
    work = torch.empty(Lb.swn_forward_work_floats(r, B, Tf), dtype=torch.float32, device=dev)
    out = torch.empty((B, n_out, Tp), dtype=torch.float32, device=dev)
    hs = torch.empty((B, L + 1, H, Tp) if want_hidden else (0,), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(Lb.swn_forward(r, _ptr(packed), _ptr(cond.contiguous()), _ptr(audio), B, Tf, _ptr(work), _ptr(out),
                                  _ptr(hs if want_hidden else None), _stream(dev)), "forward")
    return out, work, hs


stack_forward = custom_op("swn::stack_forward", mutates_args=())(stack_forward_impl)


@stack_forward.register_fake
def _(packed, cond, audio, desc, want_hidden):
    Lb, d = _lib.lib(), _desc(desc)
    _, _, L, H, n_out, _ = _geom(d)
    B, Tf = cond.shape[0], cond.shape[1]
    _, Tp = _tp(d, Tf)
    return (packed.new_empty((B, n_out, Tp)), packed.new_empty(Lb.swn_forward_work_floats(ctypes.byref(d), B, Tf)),
            packed.new_empty((B, L + 1, H, Tp) if want_hidden else (0,)))


def pack_bf16_impl(packed: torch.Tensor, desc: List[int]) -> torch.Tensor:
    """bf16 weight images of the MFMA stacks (swn_pack_bf16); raises where the geometry has no bf16 stack."""
    Lb = _lib.lib()
    d = _desc(desc)
    _need_cuda(packed, "the packed parameters")
    r = ctypes.byref(d)
    nbytes = Lb.swn_bf16_weight_bytes(r)
    if nbytes == 0:
        raise RuntimeError("bf16 stack kernels are built for the BL6-class and the H%64==0 Laplace geometries only")
    w = torch.empty(nbytes, dtype=torch.uint8, device=packed.device)
    with torch.cuda.device(packed.device):
        _lib.check(Lb.swn_pack_bf16(r, _ptr(packed), _ptr(w), _stream(packed.device)), "pack_bf16")
    return w


pack_bf16 = custom_op("swn::pack_bf16", mutates_args=())(pack_bf16_impl)


@pack_bf16.register_fake
def _(packed, desc):
    d = _desc(desc)
    return packed.new_empty(_lib.lib().swn_bf16_weight_bytes(ctypes.byref(d)), dtype=torch.uint8)


def stack_forward_bf16_impl(packed: torch.Tensor, wbf16: torch.Tensor, cond: torch.Tensor, audio: torch.Tensor,
                       desc: List[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """bf16 MFMA variant of stack_forward: raw (B, n_out, Tp) fp32 and the bf16 work buffer (swn_forward_bf16)."""
    Lb = _lib.lib()
    d = _desc(desc)
    _need_cuda(packed, "the packed parameters")
    dev = packed.device
    soft, seg, _, _, n_out, _ = _geom(d)
    B, Tf = cond.shape[0], cond.shape[1]
    T, Tp = _tp(d, Tf)
    audio = audio.to(dev, torch.int32 if soft else torch.float32).contiguous()
    if audio.numel() != B * (T - seg):
        raise RuntimeError("audio has the wrong size")
    r = ctypes.byref(d)
    work = torch.empty(Lb.swn_forward_bf16_work_bytes(r, B, Tf), dtype=torch.uint8, device=dev)
    out = torch.empty((B, n_out, Tp), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(Lb.swn_forward_bf16(r, _ptr(packed), _ptr(wbf16), _ptr(cond.contiguous()), _ptr(audio), B, Tf,
                                       _ptr(work), _ptr(out), _stream(dev)), "forward_bf16")
    return out, work


stack_forward_bf16 = custom_op("swn::stack_forward_bf16", mutates_args=())(stack_forward_bf16_impl)


@stack_forward_bf16.register_fake
def _(packed, wbf16, cond, audio, desc):
    Lb, d = _lib.lib(), _desc(desc)
    _, _, _, _, n_out, _ = _geom(d)
    B, Tf = cond.shape[0], cond.shape[1]
    _, Tp = _tp(d, Tf)
    return (packed.new_empty((B, n_out, Tp)),
            packed.new_empty(Lb.swn_forward_bf16_work_bytes(ctypes.byref(d), B, Tf), dtype=torch.uint8))


# ------------------------------------------------------------------------------------------ Laplace head
def laplace_head_impl(raw: torch.Tensor, desc: List[int], clip: bool
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """raw (B, n_out, Tp) -> mu, b, logb, a, b_clip, logb_clip (time-major; empty where not produced), below_floor."""
    Lb = _lib.lib()
    d = _desc(desc)
    _need_cuda(raw, "the stack outputs")
    dev = raw.device
    B, _, Tp = raw.shape
    seg, lpc = d.seg, d.lpc
    mk = lambda w, on=True: torch.empty((B, Tp, w) if on else (0,), dtype=torch.float32, device=dev)
    mu, b, logb = mk(seg), mk(seg), mk(seg)
    a = mk(lpc, lpc > 0)
    bc, lc = mk(seg, clip), mk(seg, clip)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.check(Lb.swn_laplace_head(ctypes.byref(d), _ptr(raw.contiguous()), B, Tp, _ptr(mu), _ptr(b), _ptr(logb),
                                       _ptr(a if lpc > 0 else None), _ptr(bc if clip else None),
                                       _ptr(lc if clip else None), _ptr(flag), _stream(dev)), "laplace_head")
    return mu, b, logb, a, bc, lc, flag


laplace_head = custom_op("swn::laplace_head", mutates_args=())(laplace_head_impl)


@laplace_head.register_fake
def _(raw, desc, clip):
    d = _desc(desc)
    B, _, Tp = raw.shape
    mk = lambda w, on=True: raw.new_empty((B, Tp, w) if on else (0,))
    return (mk(d.seg), mk(d.seg), mk(d.seg), mk(d.lpc, d.lpc > 0), mk(d.seg, clip), mk(d.seg, clip),
            raw.new_empty(1, dtype=torch.int32))


def laplace_head_backward_impl(raw: torch.Tensor, gmu: Optional[torch.Tensor], gb: Optional[torch.Tensor],
                          glogb: Optional[torch.Tensor], ga: Optional[torch.Tensor], gb_clip: Optional[torch.Tensor],
                          glogb_clip: Optional[torch.Tensor], desc: List[int]) -> torch.Tensor:
    Lb = _lib.lib()
    d = _desc(desc)
    dev = raw.device
    B, _, Tp = raw.shape
    c = lambda t: None if t is None else t.to(dev, torch.float32).contiguous()
    gmu, gb, glogb, ga, gb_clip, glogb_clip = c(gmu), c(gb), c(glogb), c(ga), c(gb_clip), c(glogb_clip)
    raw = raw.contiguous()
    graw = torch.empty_like(raw)
    with _on(dev):
        _lib.check(Lb.swn_laplace_head_backward(ctypes.byref(d), _ptr(raw), B, Tp, _ptr(gmu), _ptr(gb), _ptr(glogb),
                                                _ptr(ga), _ptr(gb_clip), _ptr(glogb_clip), _ptr(graw), _stream(dev)),
                   "laplace_head_backward")
    return graw


laplace_head_backward = custom_op("swn::laplace_head_backward", mutates_args=())(laplace_head_backward_impl)


@laplace_head_backward.register_fake
def _(raw, gmu, gb, glogb, ga, gb_clip, glogb_clip, desc):
    return torch.empty_like(raw)


# ------------------------------------------------------------------------------------------ backward of the stack
def stack_backward_impl(packed: torch.Tensor, aux: torch.Tensor, cond: torch.Tensor, fe_work: torch.Tensor,
                   audio: torch.Tensor, fwd_work: torch.Tensor, grad_raw: torch.Tensor, desc: List[int],
                   precision: int = 0) -> torch.Tensor:
    """gradient of the loss wrt the packed parameter buffer given d loss / d raw (swn_backward); precision =
    SWN_PRECISION_FP32 (0, parity) | SWN_PRECISION_BF16 (1, bf16 operands / fp32 accumulation)."""
    Lb = _lib.lib()
    d = _desc(desc)
    dev = packed.device
    B, Tf = cond.shape[0], cond.shape[1]
    r = ctypes.byref(d)
    grad_raw = grad_raw.to(dev, torch.float32).contiguous()
    work = torch.empty(Lb.swn_backward_work_floats(r, B, Tf), dtype=torch.float32, device=dev)
    gp = torch.empty_like(packed)
    with _on(dev):
        _lib.check(Lb.swn_backward(r, _ptr(packed), _ptr(aux), _ptr(cond), _ptr(fe_work), _ptr(audio), _ptr(fwd_work),
                                   _ptr(None), _ptr(grad_raw), B, Tf, _ptr(work), _ptr(gp), int(precision), _stream(dev)),
                   "backward")
    return gp


stack_backward = custom_op("swn::stack_backward", mutates_args=())(stack_backward_impl)


@stack_backward.register_fake
def _(packed, aux, cond, fe_work, audio, fwd_work, grad_raw, desc, precision=0):
    return torch.empty_like(packed)


def stack_backward_bf16_impl(packed: torch.Tensor, aux: torch.Tensor, cond: torch.Tensor, fe_work: torch.Tensor,
                        audio: torch.Tensor, work_bf16: torch.Tensor, grad_raw: torch.Tensor,
                        desc: List[int]) -> torch.Tensor:
    """stack_backward after a bf16 forward of the BL6 class with the sample-rate part fused (swn_backward_bf16): reads
    the bf16 work buffer of the forward directly; raises where swn_backward_bf16_work_floats() is 0 (callers check
    `backward_bf16_supported` first)."""
    Lb = _lib.lib()
    d = _desc(desc)
    dev = packed.device
    B, Tf = cond.shape[0], cond.shape[1]
    r = ctypes.byref(d)
    n = Lb.swn_backward_bf16_work_floats(r, B, Tf)
    if n == 0:
        raise RuntimeError("swn_backward_bf16 does not cover this geometry / size")
    grad_raw = grad_raw.to(dev, torch.float32).contiguous()
    work = torch.empty(n, dtype=torch.float32, device=dev)
    gp = torch.empty_like(packed)
    with _on(dev):
        _lib.check(Lb.swn_backward_bf16(r, _ptr(packed), _ptr(aux), _ptr(cond), _ptr(fe_work), _ptr(audio), _ptr(None),
                                        _ptr(work_bf16), _ptr(grad_raw), B, Tf, _ptr(work), _ptr(gp), _stream(dev)),
                   "backward_bf16")
    return gp


stack_backward_bf16 = custom_op("swn::stack_backward_bf16", mutates_args=())(stack_backward_bf16_impl)


@stack_backward_bf16.register_fake
def _(packed, aux, cond, fe_work, audio, work_bf16, grad_raw, desc):
    return torch.empty_like(packed)


def backward_bf16_supported(desc: List[int], batch: int, n_frames: int) -> bool:
    return _lib.lib().swn_backward_bf16_work_floats(ctypes.byref(_desc(desc)), batch, n_frames) > 0


OP_NAMES = ("pack_params", "frontend", "decode", "stack_forward", "pack_bf16", "stack_forward_bf16", "laplace_head",
            "laplace_head_backward", "stack_backward", "stack_backward_bf16")
