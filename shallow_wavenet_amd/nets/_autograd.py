"""autograd glue for training: the HIP forward/backward of the stack as
torch.autograd.Functions, and the map from packed-layout gradients back to the reference's parameters.

The chain rule through the pack-time folds (csrc/swn_pack.cpp) is applied here with a handful of tiny
tensor ops on the device: bx = b_inx + b_up * sum W_inx ; cv/cc = causal (.) wav_conv ; tap-major dil_h.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from ..config import NetConfig
from ..runtime import HipNet, layout_offsets


def unfold_packed_grads(cfg: NetConfig, gp: torch.Tensor, params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """packed-layout gradient buffer -> {state_dict key: gradient tensor}."""
    y = layout_offsets(cfg)
    H, S, K, L, seg, U = cfg.H, cfg.S, cfg.K, cfg.L, (1 if cfg.kind == "softmax" else cfg.seg), cfg.U
    H2, n, A0 = 2 * H, cfg.n_aux, cfg.A0
    Hp, Sp, O1, NO = (H + 3) & ~3, (S + 3) & ~3, cfg.out1_chn, cfg.n_out
    O1p, A0p = (O1 + 3) & ~3, (A0 + 3) & ~3
    sec = lambda name, count: gp[y[name]: y[name] + count]
    g: Dict[str, torch.Tensor] = {}
    g["scale_in.weight"] = sec("scale_w", n * n).view(n, n, 1)
    g["scale_in.bias"] = sec("scale_b", n)
    k = cfg.aux_kernel_size
    for i in range(cfg.aux_dilation_size):
        cin, cout = n * k ** i, n * k ** (i + 1)
        g[f"conv_aux.conv.{i}.weight"] = sec(f"aux_w{i}", cout * cin * k).view(cout, cin, k)
        g[f"conv_aux.conv.{i}.bias"] = sec(f"aux_b{i}", cout)
    gbx = sec("bx", L * H2).view(L, H2)             # hoisted mode: d(b_inx + b_up * sum W)
    gbt = gbx + sec("bxr", L * H2).view(L, H2)      # + dropout mode: d(b_inx) directly
    gwx = sec("wx", L * seg * H2 * A0p).view(L, seg, H2, A0p)[..., :A0]            # [l][s][o][c]
    b_up = params["upsampling.conv.bias"].reshape(())
    g["upsampling.conv.weight"] = sec("wup", U).view(1, 1, 1, U)
    c2d = cfg.kind == "laplace" and cfg.aux_conv2d_flag and seg > 1
    if c2d:   # in_x was packed as W_eff = W_in @ W2f, b_eff = b_in + W_in @ b2  (csrc/swn_pack.cpp)
        w2f = params["aux_conv2d.weight"][..., 0].reshape(A0, A0 * seg)                          # [p][c*seg+s]
        b2 = params["aux_conv2d.bias"]
        weff = [params[f"in_x.{l}.weight"][:, :, 0] @ w2f for l in range(L)]
        g["aux_conv2d.weight"] = torch.zeros_like(params["aux_conv2d.weight"])
        g["aux_conv2d.bias"] = torch.zeros_like(b2)
    else:
        weff = [params[f"in_x.{l}.weight"][:, : A0 * seg, 0] for l in range(L)]
    wsum = torch.stack([w.sum(1) for w in weff])                                                  # [l][o]
    g["upsampling.conv.bias"] = (gbx * wsum).sum().reshape(1) + sec("bup", 1)
    for l in range(L):
        gw = gwx[l].permute(1, 2, 0).reshape(H2, A0 * seg) + gbx[l][:, None] * b_up        # [o][c*seg+s]
        if cfg.kind == "softmax" and cfg.audio_in_flag:         # one-hot columns A0 .. A0+Q-1 (dswnv.py:255-256)
            Q = cfg.n_quantize
            gwa = sec("wxa", L * Q * H2).view(L, Q, H2)[l].t()                              # [o][q]
            gw = torch.cat((gw, gwa), 1)
        if c2d:
            w_in = params[f"in_x.{l}.weight"][:, :, 0]
            g["aux_conv2d.weight"] += (w_in.t() @ gw).reshape(A0, A0, seg, 1)
            g["aux_conv2d.bias"] += w_in.t() @ gbt[l]
            gw = gw @ w2f.t() + gbt[l][:, None] * b2[None, :]
        g[f"in_x.{l}.weight"] = gw.unsqueeze(2)
        g[f"in_x.{l}.bias"] = gbt[l]
    gcb = sec("cb", H)
    wc = params["causal.conv.weight"]                                            # (H, Cin, K)
    if cfg.kind == "laplace":
        gcv, gcc = sec("cv", K * H).view(K, H), sec("cc", K * H).view(K, H)
        if cfg.wav_conv_flag:
            ww, wb = params["wav_conv.weight"][:, 0, 0], params["wav_conv.bias"]
            g["wav_conv.weight"] = torch.einsum("ko,oik->i", gcv, wc).view(H, 1, 1)
            g["wav_conv.bias"] = torch.einsum("ko,oik->i", gcc, wc)
            g["causal.conv.weight"] = torch.einsum("ko,i->oik", gcv, ww) + torch.einsum("ko,i->oik", gcc, wb)
        else:
            g["causal.conv.weight"] = gcv.t().reshape(H, 1, K)
    else:
        Q = cfg.n_quantize
        gct = sec("ct", K * Q * H).view(K, Q, H)
        if cfg.wav_conv_flag:
            wq = params["wav_conv.weight"][:, :, 0] + params["wav_conv.bias"][:, None]       # (H, Q): lifted one-hot
            g["causal.conv.weight"] = torch.einsum("kqo,iq->oik", gct, wq)
            g["wav_conv.weight"] = torch.einsum("kqo,oik->iq", gct, wc).unsqueeze(2)
            g["wav_conv.bias"] = torch.einsum("kqo,oik->i", gct, wc)
        else:
            g["causal.conv.weight"] = gct.permute(2, 1, 0)
    g["causal.conv.bias"] = gcb
    gwd = sec("wd", L * H2 * K * Hp).view(L, H2, K, Hp)[..., :H]
    gbd = sec("bd", L * H2).view(L, H2)
    gwsk = sec("wsk", S * L * Hp).view(S, L, Hp)[..., :H]
    gbsk = sec("bsk", S)
    for l in range(L):
        g[f"dil_h.{l}.conv.weight"] = gwd[l].permute(0, 2, 1)
        g[f"dil_h.{l}.conv.bias"] = gbd[l]
        g[f"out_skip.{l}.weight"] = gwsk[:, l].unsqueeze(2)
        g[f"out_skip.{l}.bias"] = gbsk
    g["out_1.weight"] = sec("w1", O1 * Sp).view(O1, Sp)[:, :S].unsqueeze(2)
    g["out_1.bias"] = sec("b1", O1)
    g["out_2.weight"] = sec("w2", NO * O1p).view(NO, O1p)[:, :O1].unsqueeze(2)
    g["out_2.bias"] = sec("b2", NO)
    return g


def unfold_packed_grads_device(net: HipNet, gp: torch.Tensor, params, want) -> List:
    """the same map by ONE launch (swn_unfold_grads_device) into views of one flat buffer; `want[i]` False -> None.
    Returns None where the library does not cover the geometry (aux_conv2d_flag with seg > 1)."""
    import ctypes
    from .. import _lib
    cfg = net.cfg
    if cfg.kind == "laplace" and cfg.aux_conv2d_flag and cfg.seg > 1:
        return None
    n = len(params)
    sizes = [p.numel() if w else 0 for p, w in zip(params, want)]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=gp.device)
    base, off = flat.data_ptr(), 0
    gptr, out = [], []
    for p, sz in zip(params, sizes):
        if sz:
            gptr.append(base + 4 * off)
            out.append(flat.as_strided(p.shape, p.stride(), off))        # parameters are contiguous (checked by the caller)
            off += sz
        else:
            gptr.append(None)
            out.append(None)
    pp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in params])
    gg = (ctypes.c_void_p * n)(*gptr)
    with torch.cuda.device(gp.device):
        _lib.check(net.lib.swn_unfold_grads_device(ctypes.byref(net.desc), ctypes.c_void_p(gp.data_ptr()), pp, gg, n,
                                                   ctypes.c_void_p(torch.cuda.current_stream(gp.device).cuda_stream)),
                   "unfold_grads_device")
    return out


class StackFunction(torch.autograd.Function):
    """raw = stack(aux, audio; parameters) -> (B, n_out, Tp); backward fills every parameter's grad."""

    @staticmethod
    def forward(ctx, module, aux, audio, *params):
        net: HipNet = module._engine()
        raw, saved = net.forward_train(aux, audio, drop=getattr(module, "_pending_drop", None))
        module._pending_drop = None
        ctx.net, ctx.saved, ctx.module = net, saved, module
        ctx.plist = params                             # Module.parameters() order = state_dict order (leaves: no cycle)
        return raw

    @staticmethod
    def backward(ctx, grad_raw):
        gp = ctx.net.backward(ctx.saved, grad_raw)
        want = ctx.needs_input_grad[3:]
        fp32 = all(p.dtype == torch.float32 and p.is_contiguous() for p in ctx.plist)
        out = unfold_packed_grads_device(ctx.net, gp, ctx.plist, want) if (fp32 and ctx.module.device_unfold) else None
        if out is None:                                # torch-op version (aux_conv2d_flag, or switched off for A/B tests)
            names = [k for k, _ in ctx.module.named_parameters()]
            pd = dict(zip(names, [p.detach() for p in ctx.plist]))
            grads = unfold_packed_grads(ctx.net.cfg, gp, pd)
            out = [grads[k].reshape(pd[k].shape).contiguous() if want[i] else None for i, k in enumerate(names)]
        return (None, None, None, *out)


class LaplaceHeadFunction(torch.autograd.Function):
    """(mu, b, logb, a, b_clip, logb_clip) = head(raw)   (cswnv_shift1.py:228-267)"""

    @staticmethod
    def forward(ctx, net, raw, clip):
        from .. import ops as _ops                       # the implementation, not the dispatcher (see ops.py)
        mu, b, logb, a, bc, lc, flag = _ops.laplace_head_impl(raw, net.dlist, bool(clip))
        a = a if net.cfg.lpc > 0 else None
        bc, lc = (bc, lc) if clip else (None, None)
        ctx.net = net
        ctx.save_for_backward(raw)
        ctx.mark_non_differentiable(flag)
        empty = raw.new_zeros(0)
        return mu, b, logb, (a if a is not None else empty), (bc if bc is not None else empty), \
            (lc if lc is not None else empty), flag

    @staticmethod
    def backward(ctx, gmu, gb, glogb, ga, gbc, glc, _gflag):
        (raw,) = ctx.saved_tensors
        nz = lambda t: None if t is None or t.numel() == 0 else t
        graw = ctx.net.laplace_head_backward(raw, nz(gmu), nz(gb), nz(glogb), nz(ga), nz(gbc), nz(glc))
        return None, graw, None
