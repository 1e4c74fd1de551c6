"""GPU: the whole mixed-precision BL6 training step (drop-in module forward, LaplaceLoss, backward through the fused HIP
kernels, device unfold, re-pack, Adam with capturable state) records into ONE HIP graph and replays: nothing on the path
synchronises, allocates outside torch's pool or reads host state that changes between steps.  The replayed losses must
track an eager run of the same steps (float atomics reorder sums: 1e-3 relative over six steps)."""
import pytest
import torch

from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.runtime import train_precision
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

pytestmark = pytest.mark.gpu


def _setup(cfg, B, Tf):
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True).items()})
    m.cuda().train()
    for p in m.scale_in.parameters():
        p.requires_grad = False
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4, capturable=True)
    return m, opt


def test_training_step_replays_from_a_hip_graph(gpu_ok):
    cfg = C.bl6_laplace(1, 0)
    B, Tf = 2, 12
    T = Tf * cfg.U
    Tp = T - 2 * cfg.seg + 1
    aux = torch.from_numpy(synth_aux(cfg, B, Tf)).cuda()
    g = torch.Generator().manual_seed(2)
    audio = (torch.rand(B, 1, T - cfg.seg, generator=g) * 1.8 - 0.9).cuda()
    tgt = (torch.rand(B, Tp, generator=g) * 1.8 - 0.9).cuda()

    def make_step(m, opt):
        def step():
            res = m(aux, audio)
            loss = mc.LaplaceLoss()(res[0].reshape(B, Tp), res[1].reshape(B, Tp), tgt, log_b=res[2].reshape(B, Tp), log=False)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            return loss
        return step

    with train_precision("bf16"):
        m0, o0 = _setup(cfg, B, Tf)
        eager = [float(make_step(m0, o0)().detach()) for _ in range(6)]

        m1, o1 = _setup(cfg, B, Tf)
        step = make_step(m1, o1)
        got = []
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                      # warm-up off the default stream (library attributes, allocator pool)
            for _ in range(3):
                got.append(float(step().detach()))
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = step()
        for _ in range(3):
            graph.replay()
            got.append(float(static_loss.detach()))
    # steps 0-2 eager in both, then step 3 ran once more during capture (captured work does not execute), 3-5 replayed
    assert len(got) == 6
    for a, b in zip(got, eager):
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (got, eager)
    assert got[5] < got[0]
