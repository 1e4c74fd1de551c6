"""torch.ops.swn.*: the C ABI registered as PyTorch custom ops under one torch.library namespace (SURVEY.md 8b,
north_star "calling the HIP kernels through PyTorch-ROCm custom ops")."""
import numpy as np
import pytest
import torch

from shallow_wavenet_amd import config as C
from shallow_wavenet_amd import ops
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict


def test_every_op_is_registered_with_a_schema():
    for name in ops.OP_NAMES:
        op = getattr(torch.ops.swn, name)
        schema = str(op.default._schema)
        assert schema.startswith(f"swn::{name}("), schema
    assert "Tensor? noise" in str(torch.ops.swn.decode.default._schema)
    d = ops.desc_list(C.bl6_laplace(5, 4))
    assert len(d) == 16 and d[0] == 0 and d[2] == 64 and d[10] == 5 and d[11] == 4


def test_fake_implementations_give_the_right_shapes_without_a_device():
    """shape inference (register_fake) runs on the CPU: the size queries of the C ABI are host functions."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    cfg = C.tiny("laplace", 2, 4)
    d = ops.desc_list(cfg)
    B, Tf = 3, 5
    T = Tf * cfg.U
    with FakeTensorMode():
        packed = torch.empty(100000)
        aux = torch.empty(B, cfg.n_aux, Tf)
        cond, work = torch.ops.swn.frontend(packed, aux, d)
        assert tuple(cond.shape) == (B, Tf, cfg.L * cfg.seg * 2 * cfg.H)
        out, heads, used = torch.ops.swn.decode(packed, cond, None, None, None, d, 40, 0, 1, 0, True, True)
        assert tuple(out.shape) == (B, 80) and tuple(heads.shape) == (B, 40, cfg.n_out) and tuple(used.shape) == (B, 40, 2)
        raw, _, hs = torch.ops.swn.stack_forward(packed, cond, torch.empty(B, 1, T - 2), d, True)
        assert tuple(raw.shape) == (B, cfg.n_out, T - 3) and tuple(hs.shape) == (B, cfg.L + 1, cfg.H, T - 3)
        mu, b, logb, a, bc, lc, flag = torch.ops.swn.laplace_head(raw, d, False)
        assert tuple(mu.shape) == (B, T - 3, 2) and tuple(a.shape) == (B, T - 3, 4) and bc.numel() == 0
    with pytest.raises(RuntimeError):
        torch.ops.swn.frontend(torch.zeros(10), torch.zeros(1, cfg.n_aux, 2), d)       # real CPU tensors: no CPU path


@pytest.mark.gpu
def test_ops_match_the_runtime_and_pass_opcheck(gpu_ok):
    from shallow_wavenet_amd.runtime import HipNet
    cfg = C.tiny("laplace", 2, 4)
    sd = synth_state_dict(cfg, seed=3, flavor="trained")
    d = ops.desc_list(cfg)
    tensors = [torch.from_numpy(v).cuda() for v in sd.values()]
    packed = torch.ops.swn.pack_params(tensors, d)
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    assert torch.equal(packed, net.packed)
    aux = torch.from_numpy(synth_aux(cfg, 2, 4)).cuda()
    cond, _ = torch.ops.swn.frontend(packed, aux, d)
    noise = torch.rand(2, 40, 2, generator=torch.Generator().manual_seed(1)).sub(0.5).mul(0.99).cuda()
    out, heads, used = torch.ops.swn.decode(packed, cond, noise, None, None, d, 40, 0, 0, 0, True, False)
    out2, heads2 = net.decode(aux, 40, noise, want_heads=True)
    assert torch.equal(out, out2) and torch.equal(heads, heads2) and used.numel() == 0
    audio = out[:, None, : 4 * cfg.U - cfg.seg].contiguous()
    raw, work, _ = torch.ops.swn.stack_forward(packed, cond, audio, d, False)
    assert torch.equal(raw, net.forward(aux, audio)[0])
    torch.library.opcheck(torch.ops.swn.frontend.default, (packed, aux, d), test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(torch.ops.swn.stack_forward.default, (packed, cond, audio, d, False),
                          test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(torch.ops.swn.decode.default, (packed, cond, noise, None, None, d, 40, 0, 0, 0, True, False),
                          test_utils=("test_schema", "test_faketensor"))
