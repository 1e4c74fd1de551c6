"""GPU: swn_unfold_grads_device (one launch: packed-layout gradients -> the gradient of every parameter tensor) against
the torch-op version of the same chain rule (nets/_autograd.py unfold_packed_grads) on a random packed buffer, for every
geometry class; reductions run in double on the device and in fp32 in torch: 1e-5 relative per tensor.  The one exception is
the scalar upsampling bias: per-workgroup double partial sums meet in ONE fp32 atomic each (768 of them at the run.sh
geometry, in whatever order the workgroups finish), so it carries the rounding of that many fp32 additions - up to 2e-5 of
its magnitude was seen over 30 poisoned-allocator repetitions (tools/dbg_unfold.py): 1e-4 for that element."""
import numpy as np
import pytest
import torch

from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets._autograd import unfold_packed_grads, unfold_packed_grads_device
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu

CASES = {
    "bl6_lap": C.bl6_laplace(1, 0), "bl6_lap_seg2_lpc": C.bl6_laplace(2, 4), "ref6_lap": C.ref6_laplace(1, 4),
    "bl6_softmax": C.bl6_softmax(), "tiny_lap": C.tiny("laplace", seg=2, lpc=2), "tiny_softmax": C.tiny("softmax"),
    "tiny_lap_nowav": C.tiny("laplace", wav_conv_flag=False), "tiny_softmax_nowav": C.tiny("softmax", wav_conv_flag=False),
    "tiny_softmax_audio_in": C.tiny("softmax", audio_in_flag=True),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_unfold_matches_torch_ops(gpu_ok, name):
    cfg = CASES[name]
    sd = synth_state_dict(cfg, seed=4, flavor="trained")
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    names = [k for k, _ in cfg.param_shapes()]
    params = [torch.from_numpy(np.ascontiguousarray(sd[k])).cuda() for k in names]
    gp = torch.randn(net.packed.numel(), generator=torch.Generator().manual_seed(9)).cuda()
    want = [not k.startswith("scale_in") for k in names]
    got = unfold_packed_grads_device(net, gp, params, want)
    assert got is not None
    ref = unfold_packed_grads(cfg, gp, dict(zip(names, params)))
    torch.cuda.synchronize()
    for k, w, g, p in zip(names, want, got, params):
        if not w:
            assert g is None
            continue
        r = ref[k].reshape(p.shape)
        assert g.shape == p.shape and g.is_contiguous()
        err = float((g - r).abs().max())
        tol = 1e-4 if k == "upsampling.conv.bias" else 1e-5
        assert err <= tol * max(1.0, float(r.abs().max())), (name, k, err)


def test_conv2d_geometry_keeps_the_torch_path(gpu_ok):
    cfg = C.tiny("laplace", seg=2, aux_conv2d_flag=True)
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=4, flavor="trained"), "cuda:0")
    assert unfold_packed_grads_device(net, net.packed, [], []) is None
