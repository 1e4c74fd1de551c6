"""GPU: bf16 MFMA teacher-forced stack (training-speed path, BL6 class) against the oracle.
bf16 has 8 mantissa bits: tolerance 3e-3 absolute on raw head outputs of magnitude ~5 (observed
7e-4 max, 7e-5 mean); the fp32 kernels keep the 1e-5 parity bar (test_gpu_forward_parity.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cpu_ref
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["g1_bl6_lap_s1l0_b1_trained", "g1_bl6_lap_s1l0_b1_xavier", "g1_bl6_lap_s5l4_b1_trained"])
def test_bf16_stack_tracks_the_oracle(gpu_ok, name):
    cfg, d = load_golden(name)
    sd = synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"]))
    net, P = HipNet.from_state_dict(cfg, sd, "cuda:0"), cpu_ref.as_params(sd)
    aux, audio = torch.from_numpy(d["aux"]), torch.from_numpy(d["fwd_audio"])
    raw = net.forward_bf16(aux, audio).cpu().numpy()
    ref, _ = cpu_ref.laplace_stack(cfg, P, aux, audio)
    ref = ref.numpy()
    assert raw.shape == ref.shape
    err = np.abs(raw - ref)
    assert err.max() <= 3e-3 * max(1.0, np.abs(ref).max()), (name, err.max())
    assert err.mean() <= 3e-4


def test_bf16_stack_batch_and_ragged_length(gpu_ok):
    """B=3, a length that is not a multiple of the 16-position chunk, against the fp32 kernels."""
    cfg = C.bl6_laplace(1, 0)
    sd = synth_state_dict(cfg, seed=2, flavor="trained")
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    aux = torch.from_numpy(synth_aux(cfg, 3, 5))
    T = 5 * cfg.U
    audio = torch.rand(3, 1, T - 1, generator=torch.Generator().manual_seed(1)) * 1.6 - 0.8
    r32, _ = net.forward(aux, audio)
    r16 = net.forward_bf16(aux, audio)
    assert float((r32 - r16).abs().max()) <= 3e-3 * max(1.0, float(r32.abs().max()))


def test_bf16_unsupported_geometry_is_reported(gpu_ok):
    cfg = C.tiny("laplace", 1, 0)
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=2), "cuda:0")
    with pytest.raises(RuntimeError, match="BL6-class"):
        net.forward_bf16(torch.zeros(1, cfg.n_aux, 4), torch.zeros(1, 1, 4 * cfg.U - 1))


@pytest.mark.parametrize("seg,lpc,B,Tf", [(1, 4, 8, 38), (5, 4, 8, 38), (1, 4, 3, 130)])
def test_lds_dma_gated_layer_wide_tiles(gpu_ok, seg, lpc, B, Tf):
    """`bf16g_gate8_kernel` (the LDS-DMA gated layer of the H = 192 geometry) at sizes where its launcher takes the 192-position
    tile (8 x 38 frames: 264 tiles of 128 are two rounds over the CUs, 176 tiles of 192 are one) - the small cases of the test
    below run the 128-position form - with utterance ends inside a tile, seg = 5 conditioning taps crossing frame boundaries, and a
    batch whose tile count is odd.  Outputs against the fp32 parity kernels at the usual 3e-3 of the output scale (every level of the
    stack feeds them through skip)."""
    cfg = C.ref6_laplace(seg, lpc)
    sd = synth_state_dict(cfg, seed=6, flavor="trained")
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    aux = torch.from_numpy(synth_aux(cfg, B, Tf))
    T = Tf * cfg.U
    audio = torch.rand(B, 1, T - seg, generator=torch.Generator().manual_seed(2)) * 1.6 - 0.8
    r32, _ = net.forward(aux, audio)
    r16 = net.forward_bf16(aux, audio)
    d = (r32 - r16).abs()
    assert float(d.max()) <= 3e-3 * max(1.0, float(r32.abs().max())), float(d.max())
    assert float(d.mean()) <= 3e-4 * max(1.0, float(r32.abs().max()))


@pytest.mark.parametrize("seg,lpc,B,Tf", [(1, 4, 2, 9), (5, 4, 1, 7)])
def test_bf16_gemm_stack_for_large_geometries_tracks_the_fp32_kernels(gpu_ok, seg, lpc, B, Tf):
    """reference run.sh geometry (H=192, K=7, 3x2 layers: csrc/swn_stack_bf16g.hip) against the fp32 parity kernels:
    positions before and after the receptive field, ragged tile tails (Tp is not a multiple of 128), batch > 1.
    Tolerance as for the BL6 class: 3e-3 of the output scale (bf16 operands, fp32 accumulation)."""
    cfg = C.ref6_laplace(seg, lpc)
    sd = synth_state_dict(cfg, seed=4, flavor="trained")
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    aux = torch.from_numpy(synth_aux(cfg, B, Tf))
    T = Tf * cfg.U
    audio = torch.rand(B, 1, T - seg, generator=torch.Generator().manual_seed(1)) * 1.6 - 0.8
    r32, _ = net.forward(aux, audio)
    r16 = net.forward_bf16(aux, audio)
    assert r16.shape == r32.shape
    d = (r32 - r16).abs()
    assert float(d.max()) <= 3e-3 * max(1.0, float(r32.abs().max())), float(d.max())
    assert float(d.mean()) <= 3e-4 * max(1.0, float(r32.abs().max()))


@pytest.mark.parametrize("audio_in", [False, True])
def test_bf16_gemm_stack_softmax_model(gpu_ok, audio_in):
    """run.sh softmax geometry (H=256, K=7, 3x3 layers, Q=256), optionally with the one-hot audio input of in_x:
    bf16 GEMM stack against the fp32 parity kernels (logits of magnitude ~3)."""
    import dataclasses
    cfg = dataclasses.replace(C.ref6_softmax(), audio_in_flag=audio_in)
    sd = synth_state_dict(cfg, seed=5, flavor="xavier")
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    B, Tf = 2, 8
    aux = torch.from_numpy(synth_aux(cfg, B, Tf))
    idx = torch.randint(0, cfg.n_quantize, (B, Tf * cfg.U - 1), generator=torch.Generator().manual_seed(3))
    r32, _ = net.forward(aux, idx)
    r16 = net.forward_bf16(aux, idx)
    d = (r32 - r16).abs()
    assert float(d.max()) <= 4e-3 * max(1.0, float(r32.abs().max())), float(d.max())


def test_module_opt_in_bf16_forward(gpu_ok):
    """CSWNV.bf16_forward = True routes the no-grad forward through the bf16 stack (same tuple, small differences)."""
    from shallow_wavenet_amd.nets import cswnv_shift1 as mc
    cfg = C.ref6_laplace(1, 4)
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=6, flavor="trained").items()})
    m.cuda().eval()
    aux = torch.from_numpy(synth_aux(cfg, 1, 8)).cuda()
    audio = (torch.rand(1, 1, 8 * cfg.U - 1, generator=torch.Generator().manual_seed(2)) * 1.6 - 0.8).cuda()
    with torch.no_grad():
        ref = m(aux, audio)
        m.bf16_forward = True
        out = m(aux, audio)
    assert len(out) == len(ref) == 4
    for x, y in zip(out, ref):
        assert x.shape == y.shape
        assert float((x - y).abs().max()) <= 5e-3 * max(1.0, float(y.abs().max()))
    assert not torch.equal(out[0], ref[0])


@pytest.mark.parametrize("U", [33, 48, 64, 65, 80, 96, 110, 112])
@pytest.mark.parametrize("B,Tf", [(1, 2), (3, 7), (2, 150)])
def test_depth_fused_stack_over_upsampling_factors_and_range_cuts(gpu_ok, U, B, Tf):
    """bf16_stack_fused_kernel (all six layers in one launch, LDS rings between the layers) against the fp32 parity kernels:
    3 .. 7 chunks per conditioning frame (ragged last chunk, U a multiple of 16 and not), frame 0 / frame 1 of an utterance
    (zero padding in front, the `coff` shift), utterance boundaries inside a workgroup's range, ranges of one frame (more
    workgroups than a halo is long) and of several.  Tolerance: the bf16 path's 3e-3 of the output scale."""
    import dataclasses
    cfg = dataclasses.replace(C.bl6_laplace(1, 0), upsampling_factor=U)
    sd = synth_state_dict(cfg, seed=2, flavor="trained")
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    aux = torch.from_numpy(synth_aux(cfg, B, Tf))
    T = Tf * cfg.U
    audio = torch.rand(B, 1, T - 1, generator=torch.Generator().manual_seed(1)) * 1.6 - 0.8
    r32, _ = net.forward(aux, audio)
    r16 = net.forward_bf16(aux, audio)
    d = (r32 - r16).abs()
    assert torch.isfinite(r16).all()
    assert float(d.max()) <= 3e-3 * max(1.0, float(r32.abs().max())), (U, B, Tf, float(d.max()))
    assert float(d.mean()) <= 3e-4 * max(1.0, float(r32.abs().max()))
