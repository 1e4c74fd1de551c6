"""GPU: edge cases and error behaviour of the C-ABI entry points (empty / minimal / ragged inputs,
argument validation, determinism)."""
import ctypes

import numpy as np
import pytest
import torch

from shallow_wavenet_amd import _lib, config as C
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

pytestmark = pytest.mark.gpu


def _net(cfg, seed=4):
    return HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=seed, flavor="trained"), "cuda:0")


def test_zero_steps_is_a_no_op(gpu_ok):
    cfg = C.bl6_laplace(1, 0)
    net = _net(cfg)
    out, heads = net.decode(torch.from_numpy(synth_aux(cfg, 2, 3)), 0, torch.empty(2, 0, 1), want_heads=True)
    assert out.shape == (2, 0) and heads.shape == (2, 0, 2)


def test_single_frame_single_step(gpu_ok):
    for cfg in (C.bl6_laplace(1, 0), C.tiny("laplace", 5, 4), C.tiny("softmax")):
        net = _net(cfg)
        soft = cfg.kind == "softmax"
        w = cfg.n_quantize if soft else cfg.seg
        noise = torch.full((1, 1, w), 0.25)
        out, _ = net.decode(torch.from_numpy(synth_aux(cfg, 1, 1)), 1, noise)
        assert out.shape == (1, 1 if soft else cfg.seg)
        if not soft:
            assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0


def test_more_steps_than_conditioning_is_rejected(gpu_ok):
    cfg = C.bl6_laplace(1, 0)
    net = _net(cfg)
    n = 2 * cfg.U + 1
    with pytest.raises(RuntimeError, match="bad argument"):
        net.decode(torch.from_numpy(synth_aux(cfg, 1, 2)), n, torch.zeros(1, n, 1))


def test_wrong_shapes_raise(gpu_ok):
    cfg = C.bl6_laplace(1, 0)
    net = _net(cfg)
    with pytest.raises(RuntimeError, match="channels"):
        net.frontend(torch.zeros(1, cfg.n_aux + 1, 4))
    with pytest.raises(RuntimeError, match="noise shape"):
        net.decode(torch.from_numpy(synth_aux(cfg, 1, 2)), 10, torch.zeros(1, 11, 1))
    with pytest.raises(RuntimeError, match="audio has"):
        net.forward(torch.from_numpy(synth_aux(cfg, 1, 2)), torch.zeros(1, 1, 7))
    # null pointers at the ABI level
    d = _lib.desc_from_cfg(cfg)
    rc = _lib.lib().swn_frontend(ctypes.byref(d), None, None, 1, 2, None, None, None)
    assert rc == -2


def test_decode_is_deterministic_and_batch_independent(gpu_ok):
    """no cross-utterance dependency: an utterance decodes to the same samples alone or in a batch of 5
    (same features, same noise rows) and twice in a row (bit-exact)."""
    cfg = C.bl6_laplace(1, 0)
    net = _net(cfg)
    aux = torch.from_numpy(synth_aux(cfg, 5, 3))
    n = 3 * cfg.U
    noise = torch.empty(5, n, 1).uniform_(-0.4999, 0.5, generator=torch.Generator().manual_seed(3))
    a, _ = net.decode(aux, n, noise)
    b, _ = net.decode(aux, n, noise)
    assert torch.equal(a, b)
    solo, _ = net.decode(aux[2:3], n, noise[2:3])
    assert torch.equal(solo[0], a[2])


def test_variants_agree_on_a_ragged_batch(gpu_ok):
    cfg = C.bl6_laplace(5, 4)
    net = _net(cfg)
    aux = synth_aux(cfg, 3, 4)
    aux[1, :, 3:] = 0
    aux[2, :, 2:] = 0
    n = 4 * cfg.U // cfg.seg
    noise = torch.empty(3, n, cfg.seg).uniform_(-0.4999, 0.5, generator=torch.Generator().manual_seed(5))
    g, _ = net.decode(torch.from_numpy(aux), n, noise, variant=1)
    f, _ = net.decode(torch.from_numpy(aux), n, noise, variant=2)
    assert float((g - f).abs().max()) <= 1e-5


@pytest.mark.parametrize("lpc", [0, 4])
def test_wave_specialised_decode_against_the_oracle_and_the_symmetric_kernel(gpu_ok, lpc):
    """single-sample Laplace nets of the BL6 class decode on csrc/swn_decode_bl6w.hip (variant 0 / 2): against the CPU oracle
    (1e-5, host-drawn noise, three utterances over a frame boundary and a ragged tail), against the symmetric kernel (variant 6),
    teacher-forced inputs and the heads included; lpc = 4 has no reference-generated BL6 fixture of its own"""
    from oracle import cpu_ref
    cfg = C.bl6_laplace(1, lpc)
    sd = synth_state_dict(cfg, seed=11, flavor="trained")
    net, P = HipNet.from_state_dict(cfg, sd, "cuda:0"), cpu_ref.as_params(sd)
    B, Tf = 3, 3
    n = Tf * cfg.U
    aux = synth_aux(cfg, B, Tf, seed=5)
    aux[2, :, 2:] = 0
    aux = torch.from_numpy(aux)
    g = torch.Generator().manual_seed(9)
    noise = cpu_ref.laplace_noise(cfg, n, B, generator=g)                     # (n, B, 1)
    want = cpu_ref.laplace_generate(cfg, P, aux, [n, n, n - 37], noise)
    nz = torch.from_numpy(noise).permute(1, 0, 2).contiguous()
    new, hn = net.decode(aux, n, nz, want_heads=True, variant=2)
    old, ho = net.decode(aux, n, nz, want_heads=True, variant=6)
    for b, m in enumerate((n, n, n - 37)):
        assert np.abs(new[b, :m].cpu().numpy() - want[b][:m]).max() <= 1e-5, (lpc, b)
    assert float((new - old).abs().max()) <= 1e-5 and float((hn - ho).abs().max()) <= 1e-5
    forced = torch.from_numpy(np.stack([np.resize(w, n) for w in want]).astype(np.float32))
    fn, hfn = net.decode(aux, n, nz, forced=forced, want_heads=True, variant=2)
    fo, hfo = net.decode(aux, n, nz, forced=forced, want_heads=True, variant=6)
    assert float((fn - fo).abs().max()) <= 1e-5 and float((hfn - hfo).abs().max()) <= 1e-5


def test_forward_minimal_length(gpu_ok):
    cfg = C.tiny("laplace", 2, 4)
    net = _net(cfg)
    raw, hs = net.forward(torch.from_numpy(synth_aux(cfg, 2, 1)), torch.zeros(2, 1, cfg.U - cfg.seg), want_hidden=True)
    assert raw.shape == (2, cfg.n_out, cfg.U - 2 * cfg.seg + 1) and torch.isfinite(raw).all()
    assert hs.shape == (2, cfg.L + 1, cfg.H, cfg.U - 2 * cfg.seg + 1)


@pytest.mark.parametrize("kind,seg,lpc", [("laplace", 2, 4), ("laplace", 1, 0), ("softmax", 1, 0)])
def test_stepped_decode_many_utterances_equals_small_batches(gpu_ok, kind, seg, lpc):
    """from 24 utterances on the stepped decode works in tiles of 8 channel pairs x 8 utterances (step_layer_tile /
    rowvec_tile: a pair's weight rows fetched once per tile, the utterances' activations staged in LDS, the eight 64-lane
    sums formed by one butterfly) instead of one workgroup per pair and utterance: the same lane-by-lane sums in the same
    order, so a 27-utterance batch (three full tiles + a ragged one of three) must equal the same utterances decoded in
    batches of 14 and 13 bit for bit."""
    cfg = C.tiny(kind, seg, lpc) if kind == "laplace" else C.tiny("softmax", wav_conv_flag=False)
    net = _net(cfg)
    B, Tf = 27, 2
    aux = torch.from_numpy(synth_aux(cfg, B, Tf))
    n = Tf * cfg.U // cfg.seg
    width = cfg.n_quantize if kind == "softmax" else cfg.seg
    g = torch.Generator().manual_seed(4)
    noise = (torch.empty(B, n, width).exponential_(1, generator=g) if kind == "softmax"
             else torch.empty(B, n, width).uniform_(-0.4999, 0.5, generator=g))
    big, hb = net.decode(aux, n, noise, variant=3, want_heads=True)
    a, ha = net.decode(aux[:14], n, noise[:14], variant=3, want_heads=True)
    b, hb2 = net.decode(aux[14:], n, noise[14:], variant=3, want_heads=True)
    assert torch.equal(big, torch.cat([a, b])) and torch.equal(hb, torch.cat([ha, hb2]))
    if kind == "laplace":
        ref, _ = net.decode(aux, n, noise, variant=1)               # the generic persistent kernel
        assert float((big - ref).abs().max()) <= 1e-5


def test_stepped_decode_many_utterances_at_the_run_sh_geometry(gpu_ok):
    """the same at REF6 (H = 192, K = 7: six float4 pieces per lane and row), 25 utterances, a few generated steps."""
    cfg = C.ref6_laplace(1, 4)
    net = _net(cfg)
    B, n = 25, 24
    aux = torch.from_numpy(synth_aux(cfg, B, 1))
    noise = torch.empty(B, n, 1).uniform_(-0.4999, 0.5, generator=torch.Generator().manual_seed(6))
    big, _ = net.decode(aux, n, noise)                               # auto: stepped decode, tiles
    a, _ = net.decode(aux[:13], n, noise[:13])
    b, _ = net.decode(aux[13:], n, noise[13:])
    assert torch.equal(big, torch.cat([a, b]))
