"""GPU: swn_pack_params_device (parameters re-laid out in HBM by one launch) against the host packer
swn_pack_params - bit-identical buffers for every geometry class - and the module cache that uses it after an
optimizer step (train_cswnv_laplace-stftcmplx_shift1.py:872-874)."""
import dataclasses

import numpy as np
import pytest
import torch

from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.runtime import HipNet, pack_parameters_device, pack_state_dict
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

pytestmark = pytest.mark.gpu

CASES = {
    "tiny_lap_s1l0": C.tiny("laplace", 1, 0),
    "tiny_lap_s5l4": C.tiny("laplace", 5, 4),
    "tiny_lap_nowav": C.tiny("laplace", 1, 0, wav_conv_flag=False),
    "tiny_lap_c2d_s2l4": C.tiny("laplace", 2, 4, aux_conv2d_flag=True),
    "tiny_softmax": C.tiny("softmax", wav_conv_flag=False),
    "tiny_softmax_wav": C.tiny("softmax", wav_conv_flag=True),
    "tiny_softmax_audioin": C.tiny("softmax", wav_conv_flag=False, audio_in_flag=True),
    "bl6_lap": C.bl6_laplace(1, 0),
    "bl6_lap_s5l4": C.bl6_laplace(5, 4),
    "bl6_softmax": C.bl6_softmax(),
    "ref6_lap_s5l4": C.ref6_laplace(5, 4),
    "ref6_softmax": C.ref6_softmax(),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_pack_is_bit_identical_to_the_host_pack(gpu_ok, name):
    cfg = CASES[name]
    sd = synth_state_dict(cfg, seed=3, flavor="xavier")
    host = pack_state_dict(cfg, sd)
    tensors = [torch.from_numpy(v).cuda() for v in sd.values()]
    dev = pack_parameters_device(cfg, tensors)
    assert dev.shape == host.shape
    assert torch.equal(dev.cpu(), host), (name, int((dev.cpu() != host).sum()))
    # reuse of an output buffer full of junk: padding must be re-zeroed
    junk = torch.full_like(dev, float("nan"))
    pack_parameters_device(cfg, tensors, out=junk)
    assert torch.equal(junk.cpu(), host)


def test_module_cache_follows_an_optimizer_step_without_a_host_round_trip(gpu_ok):
    cfg = C.tiny("laplace", 2, 4)
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=5, flavor="trained").items()})
    m.cuda().train()
    aux = torch.from_numpy(synth_aux(cfg, 2, 6)).cuda()
    audio = (torch.rand(2, 1, 6 * cfg.U - cfg.seg, generator=torch.Generator().manual_seed(1)) * 1.6 - 0.8).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    net0 = m._engine()
    res = m(aux, audio)
    loss = res[0].pow(2).mean() + res[1].mean()
    loss.backward()
    opt.step()
    net1 = m._engine()
    assert net1 is net0 and net1.packed_version == 1          # refreshed in place
    fresh = HipNet.from_state_dict(cfg, m.state_dict(), "cuda:0")   # host packer on the updated parameters
    assert torch.equal(net1.packed, fresh.packed)
    with torch.no_grad():
        a = m(aux, audio)[0]
    b = fresh.laplace_head(fresh.forward(aux, audio)[0])[0]
    assert torch.equal(a, b)
    assert not torch.equal(a, res[0].detach())                 # the step really moved the outputs
