"""GPU: mixed-precision mode of the training contractions (precision = SWN_PRECISION_BF16: bf16 operands, fp32
accumulation) against the exact-fp32 mode of the same kernels and against the reference's own gradients.

Tolerance (bf16 has 8 mantissa bits; products are summed in fp32): per parameter tensor
||g_bf16 - g_ref||_2 <= 2e-2 * ||g_ref||_2 + 1e-6, where a bf16 stack exists for the geometry (BL6 class, H % 64 == 0) the forward of the step is bf16 as well
(5e-2 there); elsewhere the forward stays fp32 and the loss is untouched."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.nets import dswnv as md
from shallow_wavenet_amd.runtime import train_precision
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

pytestmark = pytest.mark.gpu

LAP = [n for n in golden_names() if "_lap_" in n and not n.startswith(("g5_", "g6_", "g9_")) and "loss" in load_golden(n)[1]]
SMX = [n for n in golden_names() if "softmax" in n and not n.startswith(("g5_", "g6_", "g9_")) and "loss" in load_golden(n)[1]]


def _grads(model):
    return {k: p.grad.detach().double().cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}


def _close(name, got, ref, tol=2e-2, floor=1e-6):
    for k, r in ref.items():
        err = np.linalg.norm((got[k] - r).ravel())
        assert err <= tol * np.linalg.norm(r.ravel()) + floor, (name, k, err, np.linalg.norm(r.ravel()))


def test_backward_runs_in_the_mode_of_its_forward(gpu_ok):
    """the arithmetic mode is an argument of every call (ABI 3); a forward records it and its backward is issued in the
    same mode even when the `train_precision` context has ended by then - in the dropout mode the work-buffer layout
    depends on it (kept gate pre-activations), so a mismatch used to read uninitialised memory."""
    from shallow_wavenet_amd import noise as swn_noise
    from shallow_wavenet_amd.runtime import HipNet, current_precision
    cfg = C.bl6_laplace(1, 0)
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=3, flavor="trained", identity_scale_in=True), "cuda:0")
    B, Tf = 2, 5
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    audio = (torch.rand(B, 1, Tf * cfg.U - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9).cuda()
    Tp = Tf * cfg.U - 2 * cfg.seg + 1
    grad_raw = (torch.randn(B, cfg.n_out, Tp, generator=torch.Generator().manual_seed(4)) / Tp).cuda()
    for drop in (None, swn_noise.dropout_masks(cfg, B, Tf, 0.5, generator=torch.Generator().manual_seed(6))):
        with train_precision("bf16"):
            _, saved = net.forward_train(aux, audio, drop=drop)
            g_in = net.backward(saved, grad_raw).clone()
            _, saved = net.forward_train(aux, audio, drop=drop)
        assert current_precision() == 0 and saved["precision"] == 1
        g_out = net.backward(saved, grad_raw)                 # after the context ended: still the forward's mode
        assert torch.isfinite(g_out).all()
        assert float((g_out - g_in).norm()) <= 1e-4 * float(g_in.norm()) + 1e-7     # float atomics reorder sums


def test_backward_after_a_repack_raises(gpu_ok):
    """a graph built before the parameters changed must not be back-propagated against the refreshed packed buffer
    (the reference raises autograd's in-place-modification error there)."""
    cfg = C.bl6_laplace(1, 0)
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=3, flavor="trained").items()})
    m.cuda().train()
    aux = torch.from_numpy(synth_aux(cfg, 1, 3)).cuda()
    audio = (torch.rand(1, 1, 3 * cfg.U - cfg.seg) * 1.8 - 0.9).cuda()
    res = m(aux, audio)
    with torch.no_grad():
        m.out_2.bias.add_(0.5)
    m(aux, audio)                       # notices the change and re-packs in place
    with pytest.raises(RuntimeError, match="modified"):
        res[0].sum().backward()


@pytest.mark.parametrize("name", LAP)
def test_laplace_gradients_bf16_mode(gpu_ok, name):
    cfg, d = load_golden(name)
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"])).items()})
    m.cuda().train()
    tgt = torch.from_numpy(d["loss_target"]).cuda()
    out = {}
    for mode in ("fp32", "bf16"):
        for p in m.parameters():
            p.grad = None
        with train_precision(mode):
            res = m(torch.from_numpy(d["aux"]).cuda(), torch.from_numpy(d["fwd_audio"]).cuda(), do=False, clip=False)
            loss = mc.LaplaceLoss()(res[0], res[1], tgt, log_b=res[2], log=False)
            if cfg.lpc > 0:
                loss = loss + 0.1 * res[3].pow(2).mean()
            assert abs(loss.item() - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
            loss.backward()
        out[mode] = _grads(m)
    _close(name, out["bf16"], out["fp32"])
    ref = {k: d[f"grad_{k}"].astype(np.float64) for k in out["fp32"] if f"grad_{k}" in d}
    _close(name, out["bf16"], ref)
    assert any(np.abs(out["bf16"][k] - out["fp32"][k]).max() > 0 for k in out["fp32"]), "bf16 mode did not engage"


@pytest.mark.parametrize("name", SMX[:2])
def test_softmax_gradients_bf16_mode(gpu_ok, name):
    cfg, d = load_golden(name)
    m = md.DSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"])).items()})
    m.cuda().train()
    idx = torch.from_numpy(d["fwd_audio_idx"]).cuda()
    tgt = torch.from_numpy(d["loss_target"]).cuda()
    out = {}
    for mode in ("fp32", "bf16"):
        for p in m.parameters():
            p.grad = None
        with train_precision(mode):
            logits = m(md.OneHot(idx, cfg.n_quantize).transpose(1, 2), torch.from_numpy(d["aux"]).cuda())
            loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tgt.reshape(-1))
            loss.backward()
        out[mode] = _grads(m)
    _close(name, out["bf16"], out["fp32"])


@pytest.mark.parametrize("shape", ["bl6", "ref6", "ref6_wide"])
def test_full_size_gradients_bf16_mode(gpu_ok, shape):
    """BASELINE cfg4-like chunk (4 x 20 frames) at the full BL6 / run.sh geometries: ragged tiles, K=7 taps.  `ref6_wide`: a batch
    at which the LDS-DMA gated layer takes its 192-position tiles (8 x 38 frames) with the pre-activations kept for the backward."""
    cfg = C.bl6_laplace(1, 0) if shape == "bl6" else C.ref6_laplace(1, 4)
    B, Tf = (8, 38) if shape == "ref6_wide" else (3, 12)
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=3, flavor="trained", identity_scale_in=True).items()})
    m.cuda().train()
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    T = Tf * cfg.U
    audio = (torch.rand(B, 1, T - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9).cuda()
    Tp = T - 2 * cfg.seg + 1
    tgt = (torch.rand(B, Tp, generator=torch.Generator().manual_seed(3)) * 1.8 - 0.9).cuda()
    out = {}
    for mode in ("fp32", "bf16"):
        for p in m.parameters():
            p.grad = None
        with train_precision(mode):
            res = m(aux, audio)
            loss = mc.LaplaceLoss()(res[0].reshape(B, Tp), res[1].reshape(B, Tp), tgt, log_b=res[2].reshape(B, Tp), log=False)
            loss.backward()
        out[mode] = _grads(m)
    # the forward runs on the bf16 stack as well (runtime._bf16_train_forward), so the gradients carry the rounding of
    # both passes
    # (the floor covers the scalar upsampler bias, a sum of cancelling terms: 1e-3 of the largest tensor norm)
    big = max(np.linalg.norm(v.ravel()) for v in out["fp32"].values())
    _close(shape, out["bf16"], out["fp32"], tol=5e-2, floor=1e-3 * big)


@pytest.mark.parametrize("shape", ["bl6", "ref6"])
def test_bf16_forward_feeds_the_backward(gpu_ok, shape):
    """mixed-precision mode where a bf16 stack exists: swn_forward_bf16 + swn_bf16_work_to_f32 must leave in the fp32
    work buffer what swn_forward would have left (hidden states, relu(skip), relu(out_1)) up to bf16 rounding.  At the
    BL6 class the two head activations are recomputed in fp32 from the expanded hidden states."""
    from shallow_wavenet_amd.runtime import HipNet
    cfg = C.bl6_laplace(1, 0) if shape == "bl6" else C.ref6_laplace(1, 4)
    sd = synth_state_dict(cfg, seed=3, flavor="trained", identity_scale_in=True)
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    B, Tf = 2, 9
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    audio = (torch.rand(B, 1, Tf * cfg.U - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9).cuda()
    raw32, s32 = net.forward_train(aux, audio)
    net.fused_backward = False          # BL6 class: the fused backward reads the bf16 buffer itself and skips this expansion
    with train_precision("bf16"):
        raw16, s16 = net.forward_train(aux, audio)
    assert raw16.shape == raw32.shape and s16["work"].shape == s32["work"].shape
    scale = float(raw32.abs().max())
    assert float((raw16 - raw32).abs().max()) <= 5e-3 * max(1.0, scale)
    assert float((raw16 - raw32).abs().max()) > 0, "bf16 forward did not engage"
    Tp = Tf * cfg.U - 2 * cfg.seg + 1
    r64 = lambda x: (x + 63) & ~63
    n_hs, n_s1, n_r1 = B * (cfg.L + 1) * cfg.H * Tp, B * cfg.S * Tp, B * cfg.out1_chn * Tp
    o = 0
    for j, n in enumerate((n_hs, n_s1, n_r1)):    # the three sections of the work layout (the padding between them is never read)
        w32, w16 = s32["work"][o:o + n], s16["work"][o:o + n]
        assert float((w16 - w32).abs().max()) <= 2e-2 * max(1.0, float(w32.abs().max())), (shape, j)
        if j == 0 or shape == "ref6":             # expanded from bf16 storage: every value is a bf16 number
            assert torch.equal(w16, w16.to(torch.bfloat16).to(torch.float32)), (shape, j)
        o += r64(n)


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("g5_drop") and "_lap_" in n])
def test_dropout_gradients_bf16_mode(gpu_ok, name):
    """dropout mode takes the masked-operand kernels (X * mask staged into LDS) and evaluates in_x at sample rate
    through the same contraction, so in bf16 mode the forward itself carries bf16 rounding: outputs within 2e-2 abs
    of the reference's, gradients within the tensor-norm tolerance of the reference's own gradients."""
    cfg, d = load_golden(name)
    m = mc.CSWNV(**cfg.ctor_kwargs(), do_prob=float(d["drop_p"]))
    m.dropout_source = "host"            # fixture = the reference's CPU masks
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"])).items()})
    m.cuda().train()
    tgt = torch.from_numpy(d["loss_target"]).cuda()
    with train_precision("bf16"):
        torch.manual_seed(int(d["drop_seed"]))
        do = bool(int(d["do"])) if "do" in d else True
        res = m(torch.from_numpy(d["aux"]).cuda(), torch.from_numpy(d["fwd_audio"]).cuda(), do=do, clip=False)
        for i, r in enumerate(res):
            assert np.abs(r.detach().cpu().numpy() - d[f"fwd_{i}"]).max() <= 2e-2, (name, i)
        loss = mc.LaplaceLoss()(res[0], res[1], tgt, log_b=res[2], log=False)
        if cfg.lpc > 0:
            loss = loss + 0.1 * res[3].pow(2).mean()
        loss.backward()
    got = _grads(m)
    ref = {k: d[f"grad_{k}"].astype(np.float64) for k in got if f"grad_{k}" in d}
    assert ref
    # these 32-channel nets keep the exact-fp32 forward (only H % 64 == 0 nets run the dropout-mode forward on bf16
    # operands, see test_full_size_dropout_step_bf16_mode); the sample-rate in_x product and the backward are rounded: the
    # smallest tensors (norm 5e-3) sit at 3-5 %; the floor covers the scalar upsampler bias, a sum of cancelling terms of 2e-4
    _close(name, got, ref, tol=6e-2, floor=5e-5)


@pytest.mark.parametrize("shape", ["bl6", "ref6", "ref6_wide"])
def test_full_size_dropout_step_bf16_mode(gpu_ok, shape):
    """dropout mode (do_prob = 0.5, forward(do=True): how run.sh trains) at the full BL6 / run.sh geometries, same masks in
    both modes: the mixed-precision mode runs the gated layers and the wide head layers of the forward on bf16 operands
    and hands the gate pre-activations to the backward; against the fp32 mode of the same kernels, 5e-2 per tensor like
    the step without dropout."""
    cfg = C.bl6_laplace(1, 0) if shape == "bl6" else C.ref6_laplace(1, 4)
    B, Tf = (8, 38) if shape == "ref6_wide" else (3, 12)       # ref6_wide: the gated layers run on 192-position tiles
    m = mc.CSWNV(**cfg.ctor_kwargs(), do_prob=0.5)
    m.dropout_source = "host"            # the same masks for both modes
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=3, flavor="trained", identity_scale_in=True).items()})
    m.cuda().train()
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    T = Tf * cfg.U
    audio = (torch.rand(B, 1, T - cfg.seg, generator=torch.Generator().manual_seed(2)) * 1.8 - 0.9).cuda()
    Tp = T - 2 * cfg.seg + 1
    tgt = (torch.rand(B, Tp, generator=torch.Generator().manual_seed(3)) * 1.8 - 0.9).cuda()
    out, fwd = {}, {}
    for mode in ("fp32", "bf16"):
        for p in m.parameters():
            p.grad = None
        with train_precision(mode):
            torch.manual_seed(11)
            res = m(aux, audio, do=True)
            fwd[mode] = [r.detach().clone() for r in res[:3]]
            loss = mc.LaplaceLoss()(res[0].reshape(B, Tp), res[1].reshape(B, Tp), tgt, log_b=res[2].reshape(B, Tp), log=False)
            loss.backward()
        out[mode] = _grads(m)
    for a, b in zip(fwd["bf16"], fwd["fp32"]):
        assert float((a - b).abs().max()) <= 2e-2 * max(1.0, float(b.abs().max()))
    assert float((fwd["bf16"][0] - fwd["fp32"][0]).abs().max()) > 0, "the bf16 forward of the dropout mode did not engage"
    big = max(np.linalg.norm(v.ravel()) for v in out["fp32"].values())
    _close(shape, out["bf16"], out["fp32"], tol=5e-2, floor=1e-3 * big)


def test_softmax_run_sh_geometry_bf16_mode(gpu_ok):
    """DSWNV at the run.sh softmax geometry (H = 256, GEMM-stack class: int32 class indices through the bf16 forward,
    logits and cross-entropy gradients): mixed-precision mode against the fp32 mode of the same kernels."""
    cfg = C.ref6_softmax()
    B, Tf = 2, 10
    m = md.DSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=5, flavor="trained", identity_scale_in=True).items()})
    m.cuda().train()
    T = Tf * cfg.U
    idx = torch.randint(0, cfg.n_quantize, (B, T - 1), generator=torch.Generator().manual_seed(1)).cuda()
    tgt = torch.randint(0, cfg.n_quantize, (B, T - 1), generator=torch.Generator().manual_seed(2)).cuda()
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    out, lg = {}, {}
    for mode in ("fp32", "bf16"):
        for p in m.parameters():
            p.grad = None
        with train_precision(mode):
            logits = m(md.OneHot(idx, cfg.n_quantize).transpose(1, 2), aux)
            loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tgt.reshape(-1))
            loss.backward()
        out[mode], lg[mode] = _grads(m), logits.detach()
    assert float((lg["bf16"] - lg["fp32"]).abs().max()) <= 2e-2 * max(1.0, float(lg["fp32"].abs().max()))
    assert float((lg["bf16"] - lg["fp32"]).abs().max()) > 0
    big = max(np.linalg.norm(v.ravel()) for v in out["fp32"].values())
    # nine layers deep, both passes rounded, random labels: the frame-rate front end (norm 7e-3 of 0.02) sits at 5-6 %
    _close("ref6_softmax", out["bf16"], out["fp32"], tol=1e-1, floor=1e-3 * big)


def test_kept_preactivations_match_the_recompute(gpu_ok):
    """GEMM-stack class, mixed-precision mode: swn_forward_bf16_keep + swn_backward_keep (no recompute GEMM) against
    swn_forward_bf16 + swn_backward (recompute from the expanded hidden states) - the same bf16 operands either way, so the
    packed gradients agree to summation order."""
    from shallow_wavenet_amd.runtime import HipNet
    for cfg in (C.ref6_laplace(1, 4), C.ref6_softmax()):
        net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=3, flavor="trained", identity_scale_in=True), "cuda:0")
        B, Tf = 2, 9
        soft = cfg.kind == "softmax"
        aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
        T = Tf * cfg.U
        g = torch.Generator().manual_seed(2)
        if soft:
            audio = torch.randint(0, cfg.n_quantize, (B, T - 1), generator=g).cuda()
            Tp = T - 1
        else:
            audio = (torch.rand(B, 1, T - cfg.seg, generator=g) * 1.8 - 0.9).cuda()
            Tp = T - 2 * cfg.seg + 1
        grad_raw = (torch.randn(B, cfg.n_out, Tp, generator=g) / Tp).cuda()
        got = {}
        with train_precision("bf16"):
            for keep in (True, False):
                net.keep_preactivations = keep
                raw, saved = net.forward_train(aux, audio)
                assert (saved.get("a_keep") is not None) == keep
                got[keep] = (raw.clone(), net.backward(saved, grad_raw).clone())
        assert torch.equal(got[True][0], got[False][0])
        a, b = got[True][1].double(), got[False][1].double()
        assert float((a - b).norm()) <= 2e-3 * float(b.norm()), (cfg.kind, float((a - b).norm()), float(b.norm()))


def test_gemm_stack_dropout_step_softmax_audio_in(gpu_ok):
    """the dropout-mode forward of the mixed-precision mode on the bf16 GEMM stack (swn_drop_g16: H % 64 == 0 nets) with everything
    the run.sh fixtures do not switch on at once: a softmax net with the one-hot audio columns of in_x (`audio_in_flag`), a mask
    between layers (dilation_repeat 2), an odd sequence length (the G4 layout's ragged last block, the bf16 rows' ragged last
    octet), three utterances.  Same host-drawn masks in both arithmetic modes; logits within 2e-2 of scale, every gradient
    within 1e-1 of the fp32 mode's (the softmax tolerance of the teacher-forced fixtures)."""
    cfg = C.NetConfig(kind="softmax", n_aux=10, hid_chn=64, skip_chn=64, dilation_depth=2, dilation_repeat=2, kernel_size=3,
                      upsampling_factor=37, n_quantize=64, wav_conv_flag=False, audio_in_flag=True)
    B, Tf = 3, 9                                                      # T = 333, Tp = 332 positions (>= 256: the bf16-copy kernels)
    m = md.DSWNV(**cfg.ctor_kwargs(), do_prob=0.5)
    m.dropout_source = "host"
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=5, flavor="trained").items()})
    m.cuda().train()
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    T = Tf * cfg.U
    g = torch.Generator().manual_seed(4)
    idx = torch.randint(0, cfg.n_quantize, (B, T - 1), generator=g).cuda()
    tgt = torch.randint(0, cfg.n_quantize, (B, T - 1), generator=g).cuda()
    out, fwd = {}, {}
    for mode in ("fp32", "bf16"):
        for p in m.parameters():
            p.grad = None
        with train_precision(mode):
            torch.manual_seed(13)
            logits = m(md.OneHot(idx, cfg.n_quantize).transpose(1, 2), aux, do=True)
            fwd[mode] = logits.detach().clone()
            torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), tgt.reshape(-1)).backward()
        out[mode] = _grads(m)
    assert float((fwd["bf16"] - fwd["fp32"]).abs().max()) <= 2e-2 * max(1.0, float(fwd["fp32"].abs().max()))
    assert float((fwd["bf16"] - fwd["fp32"]).abs().max()) > 0, "the bf16 forward did not engage"
    _close("g16_softmax_audio_in", out["bf16"], {k: v.astype(np.float64) for k, v in out["fp32"].items()}, tol=1e-1, floor=5e-5)


@pytest.mark.parametrize("geom", ["bl6", "ref6"])
def test_loss_curves_of_the_arithmetic_modes_track_each_other(gpu_ok, geom):
    """Twenty Adam steps on one fixed synthetic batch (tools/train_sanity.py in short): the mixed-precision mode - through the
    fused backward where the geometry has one, and through the generic chain - must end within 2 % of the loss the exact-fp32
    mode reaches, and every curve must descend.  """
    cfg = C.bl6_laplace(1, 0) if geom == "bl6" else C.ref6_laplace(1, 4)
    B, Tf, N = (4, 12, 20) if geom == "bl6" else (2, 8, 20)
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    T = Tf * cfg.U
    Tp = T - 2 * cfg.seg + 1
    g = torch.Generator().manual_seed(2)
    audio = (torch.rand(B, 1, T - cfg.seg, generator=g) * 0.6 - 0.3).cuda()
    tgt = (torch.rand(B, Tp, generator=g) * 0.6 - 0.3).cuda()
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True).items()}
    curves = {}
    for name, mode, fused in (("fp32", "fp32", True), ("bf16 chain", "bf16", False), ("bf16 fused", "bf16", True)):
        m = mc.CSWNV(**cfg.ctor_kwargs())
        m.load_state_dict(sd)
        m.cuda().train()
        for p in m.scale_in.parameters():
            p.requires_grad = False
        # (lr 2e-4: at 1e-3 the REF6 loss falls 20x in these 20 steps and the trajectories of the modes - and of two runs of one mode, through
        #  the float atomics of the weight gradients - spread to within 10 % of the tolerance below; here they agree to 3 % of the final loss)
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=2e-4)
        losses = []
        with train_precision(mode):
            for _ in range(N):
                m._engine().fused_backward = fused
                res = m(aux, audio)
                loss = mc.LaplaceLoss()(res[0].reshape(B, Tp), res[1].reshape(B, Tp), tgt, log_b=res[2].reshape(B, Tp), log=False)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                losses.append(float(loss))
        curves[name] = losses
    ref = curves["fp32"]
    assert ref[-1] < ref[0], ref
    for name in ("bf16 chain", "bf16 fused"):
        c = curves[name]
        assert c[-1] < c[0], (name, c)
        assert abs(c[-1] - ref[-1]) <= 2e-2 * max(abs(ref[-1]), 1e-3) + 2e-2 * abs(ref[0] - ref[-1]), (name, c[-1], ref[-1], ref[0])
