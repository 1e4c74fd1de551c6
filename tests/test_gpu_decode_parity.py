"""GPU parity tests of the decode hot path (through the C ABI) against the oracle and the
golden fixtures generated from the reference.

Tolerances (north_star / SURVEY.md 8c):
  * Laplace free-running samples  <= 1e-5 abs over the fixture length
  * Laplace teacher-forced heads  <= 2e-6 abs (mu, pre-sigmoid scale, LP coefficients)
  * softmax indices bit-exact (fixtures have top-2 margins >= 1e-4), logits <= 2e-5
"""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import cpu_ref
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu

TOL_FREE = 1e-5
TOL_TF = 2e-6
BIG_STEPS = 400          # REF6 fixtures: first steps only (generic kernel streams 15-24 MB/step)

LAP = [n for n in golden_names() if "_lap_" in n and not n.startswith(("g5_", "g6_", "g9_"))]
SMX = [n for n in golden_names() if "softmax" in n and not n.startswith(("g5_", "g6_", "g9_"))]


def _net(cfg, d):
    sd = synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"]))
    return HipNet.from_state_dict(cfg, sd, "cuda:0"), cpu_ref.as_params(sd)


def _variants(cfg):
    """1 generic persistent, 3 stepped multi-launch, 0 auto (BL6 fast kernels / stepped for REF6); BL6-class nets also 6 = the
    symmetric BL6 kernel (auto takes the wave-specialised one for the single-sample Laplace nets)"""
    bl6 = cfg.H == 64 and cfg.kernel_size == 2 and cfg.dilation_depth == 6 and cfg.dilation_repeat == 1
    return [1, 3, 0] + ([6] if bl6 else [])


@pytest.mark.parametrize("name", LAP)
def test_laplace_free_running_matches_reference(gpu_ok, name):
    cfg, d = load_golden(name)
    net, _ = _net(cfg, d)
    n_steps = d["noise"].shape[0]
    if name.startswith("g2_"):
        n_steps = min(n_steps, BIG_STEPS // cfg.seg)
    noise = torch.from_numpy(d["noise"][:n_steps]).permute(1, 0, 2).contiguous()
    for variant in _variants(cfg):
        out, heads = net.decode(torch.from_numpy(d["aux"]), n_steps, noise, want_heads=True, variant=variant)
        out, heads = out.cpu().numpy(), heads.cpu().numpy()
        ref_heads = np.transpose(d["heads"][:n_steps], (1, 0, 2))
        assert np.abs(heads - ref_heads).max() <= TOL_FREE, (name, variant)
        for b, n in enumerate(d["n_samples"]):
            n = min(int(n), n_steps * cfg.seg)
            assert np.abs(out[b, :n] - d[f"samples_{b}"][:n]).max() <= TOL_FREE, (name, variant)


@pytest.mark.parametrize("name", [n for n in LAP if not n.startswith("g2_")])
def test_laplace_teacher_forced_matches_oracle(gpu_ok, name):
    cfg, d = load_golden(name)
    net, P = _net(cfg, d)
    B = d["aux"].shape[0]
    n_steps = d["noise"].shape[0]
    full, heads_ref = cpu_ref.laplace_generate(cfg, P, torch.from_numpy(d["aux"]),
                                               [n_steps * cfg.seg] * B, d["noise"], return_heads=True)
    forced = torch.from_numpy(np.stack(full))
    noise = torch.from_numpy(d["noise"]).permute(1, 0, 2).contiguous()
    for variant in _variants(cfg):
        out, heads = net.decode(torch.from_numpy(d["aux"]), n_steps, noise, forced=forced,
                                want_heads=True, variant=variant)
        got, ref = heads.cpu().numpy(), np.transpose(heads_ref, (1, 0, 2))
        seg = cfg.seg
        sig = lambda y: 1.0 / (1.0 + np.exp(-y.astype(np.float64)))
        assert np.abs(got[..., :seg] - ref[..., :seg]).max() <= TOL_TF, (name, variant)            # mu
        assert np.abs(sig(got[..., seg:2 * seg]) - sig(ref[..., seg:2 * seg])).max() <= 1e-6       # b
        assert np.abs(got[..., 2 * seg:] - ref[..., 2 * seg:]).max(initial=0.0) <= TOL_TF          # a
        assert np.abs(got - ref).max() <= 1e-5
        assert np.abs(out.cpu().numpy() - np.stack(full)).max() <= TOL_TF, (name, variant)


@pytest.mark.parametrize("name", SMX)
def test_softmax_free_running_bit_exact(gpu_ok, name):
    cfg, d = load_golden(name)
    net, _ = _net(cfg, d)
    B = d["aux"].shape[0]
    n_steps = int(d["n_samples"].max())
    if "q" in d:
        q = d["q"]
    else:
        g = torch.Generator().manual_seed(int(d["noise_seed"]))
        q = cpu_ref.softmax_noise(cfg, n_steps, B, generator=g)
    if name.startswith("g2_"):
        n_steps = min(n_steps, BIG_STEPS)
    noise = torch.from_numpy(q[:n_steps]).permute(1, 0, 2).contiguous()
    st = int(d["head_stride"])
    for variant in _variants(cfg):
        out, heads = net.decode(torch.from_numpy(d["aux"]), n_steps, noise, want_heads=True, variant=variant)
        out, heads = out.cpu().numpy(), heads.cpu().numpy()
        ref_heads = np.transpose(d["heads"], (1, 0, 2))
        got = heads[:, :n_steps:st]
        assert np.abs(got - ref_heads[:, : got.shape[1]]).max() <= 2e-5, (name, variant)
        for b, n in enumerate(d["n_samples"]):
            n = min(int(n), n_steps)
            assert np.array_equal(out[b, :n], d[f"samples_{b}"][:n]), (name, variant)


def test_frontend_matches_oracle(gpu_ok):
    for name in ("g0_tiny_lap_s5l4_trained", "g1_bl6_lap_s1l0_b3_trained", "g0_tiny_softmax"):
        cfg, d = load_golden(name)
        net, P = _net(cfg, d)
        aux = torch.from_numpy(d["aux"])
        cond = net.frontend(aux).cpu().double()
        c = cpu_ref.frontend(cfg, P, aux).double()                       # B, A0, Tf
        seg = 1 if cfg.kind == "softmax" else cfg.seg
        ref = []
        for l in range(cfg.L):
            w = P[f"in_x.{l}.weight"][:, : cfg.A0 * seg, 0].double()     # 2H, A0*seg
            w = w.reshape(2 * cfg.H, cfg.A0, seg)
            ref.append(torch.einsum("ocs,bcf->bfso", w, c))              # B,Tf,seg,2H
        ref = torch.stack(ref, 2).reshape(cond.shape)                     # B,Tf,L,seg,2H
        scale = ref.abs().max().item()
        assert (cond - ref).abs().max().item() <= 2e-6 * max(1.0, scale), name
