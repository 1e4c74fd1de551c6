"""GPU: sampling noise drawn INSIDE the decode kernels (swn_decode_io.noise_dev == NULL, csrc/swn_noise.hpp) and the
caller-supplied seed waveform.

Protocol (SURVEY.md 8c "given the same noise"): the kernels dump every deviate they used; the CPU oracle replays the run
from that dump and must produce the same samples (Laplace <= 1e-5) / the same indices (softmax, bit-exact).  The dump
itself is pinned against the numpy restatement of the generator (Philox4x32-10, Random123 known answers in
tests/test_oracle_golden.py).  All decode variants draw the same stream: results do not depend on the kernel,
on batch composition or on sharding."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.nets import dswnv as md
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

pytestmark = pytest.mark.gpu
SEED = 0x5EED0123456789AB


def _net(cfg, seed=7, flavor="trained"):
    sd = synth_state_dict(cfg, seed=seed, flavor=flavor)
    return HipNet.from_state_dict(cfg, sd, "cuda:0"), cpu_ref.as_params(sd)


@pytest.mark.parametrize("cfgname,variants", [("tiny_s1l0", (1, 3)), ("tiny_s5l4", (1, 3)), ("tiny_s2l4", (1, 3)),
                                              ("bl6_s1l0", (2, 6, 1)), ("bl6_s1l4", (2, 6, 1)), ("bl6_s5l4", (2, 1))])
def test_laplace_device_noise_replays_in_the_oracle(gpu_ok, cfgname, variants):
    cfg = {"tiny_s1l0": C.tiny("laplace", 1, 0), "tiny_s5l4": C.tiny("laplace", 5, 4), "tiny_s2l4": C.tiny("laplace", 2, 4),
           "bl6_s1l0": C.bl6_laplace(1, 0), "bl6_s1l4": C.bl6_laplace(1, 4), "bl6_s5l4": C.bl6_laplace(5, 4)}[cfgname]
    # (BL6 single-sample nets: 2 = the wave-specialised kernel in its extended instantiation, 6 = the symmetric kernel)
    net, P = _net(cfg)
    B, Tf = 3, 4
    n_steps = Tf * cfg.U // cfg.seg
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=3))
    ref_noise = cpu_ref.device_noise("laplace", SEED, 5, B, n_steps, cfg.seg)
    outs = []
    for v in variants:
        out, _, used = net.decode(aux, n_steps, None, variant=v, rng_seed=SEED, rng_utt0=5, want_noise=True)
        used = used.cpu().numpy()
        assert np.array_equal(used, ref_noise), (cfgname, v)                 # the stream is the documented generator
        outs.append(out.cpu().numpy())
    want = cpu_ref.laplace_generate(cfg, P, aux, [n_steps * cfg.seg] * B, np.ascontiguousarray(ref_noise.transpose(1, 0, 2)))
    for v, o in zip(variants, outs):
        for b in range(B):
            assert np.abs(o[b] - want[b]).max() <= 1e-5, (cfgname, v, b)
    # sharding / batching independence: utterance 2 decoded alone as global utterance 5 + 2
    solo, _ = net.decode(aux[2:3], n_steps, None, variant=variants[0], rng_seed=SEED, rng_utt0=7)
    assert np.array_equal(solo.cpu().numpy()[0], outs[0][2])
    other, _ = net.decode(aux, n_steps, None, variant=variants[0], rng_seed=SEED + 1, rng_utt0=5)
    assert not np.array_equal(other.cpu().numpy(), outs[0])


@pytest.mark.parametrize("cfgname,variants", [("tiny", (1, 3)), ("tiny_wav", (1, 3)), ("bl6", (2, 1))])
def test_softmax_device_noise_replays_in_the_oracle_bit_exact(gpu_ok, cfgname, variants):
    cfg = {"tiny": C.tiny("softmax", wav_conv_flag=False), "tiny_wav": C.tiny("softmax", wav_conv_flag=True),
           "bl6": C.bl6_softmax()}[cfgname]
    net, P = _net(cfg, flavor="xavier")
    B, Tf = 2, 3
    n_steps = Tf * cfg.U
    aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=4))
    ref_q = cpu_ref.device_noise("softmax", SEED, 0, B, n_steps, cfg.n_quantize)
    first = None
    for v in variants:
        out, heads, used = net.decode(aux, n_steps, None, variant=v, rng_seed=SEED, want_heads=True, want_noise=True)
        q = used.cpu().numpy()
        assert q.min() > 0 and np.abs(q / ref_q - 1.0).max() <= 2e-6, (cfgname, v)   # device logf vs libm
        if first is None:
            first = (out.cpu().numpy(), q)
        else:
            assert np.array_equal(q, first[1]) and np.array_equal(out.cpu().numpy(), first[0]), (cfgname, v)
    idx, q = first
    want, _, margins = cpu_ref.softmax_generate(cfg, P, aux, [n_steps] * B, np.ascontiguousarray(q.transpose(1, 0, 2)),
                                                return_heads=True)
    for b in range(B):
        if not np.array_equal(idx[b], want[b]):
            # only a near-tie of the oracle's own p/q ranking may flip an index; everything before it must agree
            t = int(np.nonzero(idx[b] != want[b])[0][0])
            assert float(np.asarray(margins)[t, b]) < 1e-4, (cfgname, b, t)
            assert np.array_equal(idx[b][:t], want[b][:t])
    assert abs(float(q.mean()) - 1.0) < 0.05 and len(np.unique(idx)) > 8


def test_nonzero_seed_waveform_matches_the_oracle(gpu_ok):
    """batch_fast_generate(audio != 0): the seed samples enter the first causal window and the LP buffer
    (cswnv_shift1.py:300-334); every decode variant, seg = 1 and seg = 5."""
    for cfg, variants in ((C.tiny("laplace", 5, 4), (1, 3)), (C.tiny("laplace", 1, 4), (1, 3)), (C.bl6_laplace(5, 4), (2, 1)),
                          (C.bl6_laplace(1, 0), (2,))):
        net, P = _net(cfg)
        B, Tf = 2, 3
        n_steps = Tf * cfg.U // cfg.seg
        aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=6))
        seed = torch.tensor(np.random.Generator(np.random.PCG64(2)).uniform(-0.8, 0.8, (B, cfg.seg)).astype(np.float32))
        noise = cpu_ref.laplace_noise(cfg, n_steps, B, generator=torch.Generator().manual_seed(9))
        want = cpu_ref.laplace_generate(cfg, P, aux, [n_steps * cfg.seg] * B, noise, seed=seed)
        zero = cpu_ref.laplace_generate(cfg, P, aux, [n_steps * cfg.seg] * B, noise)
        assert np.abs(want[0] - zero[0]).max() > 1e-4                         # the seed matters
        for v in variants:
            out, _ = net.decode(aux, n_steps, torch.from_numpy(noise).permute(1, 0, 2).contiguous(), variant=v, seed=seed)
            for b in range(B):
                assert np.abs(out[b].cpu().numpy() - want[b]).max() <= 1e-5, (cfg.seg, cfg.lpc, v, b)


def test_nonzero_seed_class_softmax(gpu_ok):
    for cfg, variants in ((C.tiny("softmax", wav_conv_flag=False), (1, 3)), (C.bl6_softmax(), (2, 1))):
        net, P = _net(cfg, flavor="xavier")
        B, Tf = 2, 2
        n_steps = Tf * cfg.U
        aux = torch.from_numpy(synth_aux(cfg, B, Tf, seed=6))
        seed = torch.tensor([17, 201])
        q = cpu_ref.softmax_noise(cfg, n_steps, B, generator=torch.Generator().manual_seed(11))
        want = cpu_ref.softmax_generate(cfg, P, aux, [n_steps] * B, q, seed=seed)
        for v in variants:
            out, _ = net.decode(aux, n_steps, torch.from_numpy(q).permute(1, 0, 2).contiguous(), variant=v, seed=seed)
            for b in range(B):
                assert np.array_equal(out[b].cpu().numpy(), want[b]), (v, b)


def test_modules_accept_seed_and_device_noise(gpu_ok):
    cfg = C.tiny("laplace", 2, 4)
    sd = synth_state_dict(cfg, seed=7, flavor="trained")
    m = mc.CSWNV(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.cuda().eval()
    aux = torch.from_numpy(synth_aux(cfg, 2, 3, seed=6)).cuda()
    audio = torch.tensor([[0.3, -0.2], [0.1, 0.4]]).cuda()
    n = [3 * cfg.U, 2 * cfg.U]
    torch.manual_seed(4)
    got = m.batch_fast_generate(audio, aux, n)                                 # default: host stream
    g = torch.Generator().manual_seed(4)
    noise = cpu_ref.laplace_noise(cfg, n[0] // cfg.seg, 2, generator=g)
    want = cpu_ref.laplace_generate(cfg, cpu_ref.as_params(sd), aux.cpu(), n, noise, seed=audio.cpu())
    for b in range(2):
        assert got[b].shape == (n[b],) and np.abs(got[b] - want[b]).max() <= 1e-5
    m.noise_source = "device"
    torch.manual_seed(4)
    a = m.batch_fast_generate(audio, aux, n)
    torch.manual_seed(4)
    b2 = m.batch_fast_generate(audio, aux, n)
    assert np.array_equal(a[0], b2[0]) and not np.array_equal(a[0], got[0])    # reproducible, but another stream
    s = md.DSWNV(**C.tiny("softmax", wav_conv_flag=False).ctor_kwargs()).cuda().eval()
    auxs = torch.from_numpy(synth_aux(s._cfg, 1, 2, seed=6)).cuda()
    torch.manual_seed(5)
    i1 = s.batch_fast_generate(torch.tensor([[77]]).cuda(), auxs, [2 * s._cfg.U])   # default: drawn on the device
    torch.manual_seed(5)
    i2 = s.batch_fast_generate(torch.tensor([[77]]).cuda(), auxs, [2 * s._cfg.U])
    assert i1[0].dtype == np.int64 and np.array_equal(i1[0], i2[0])
    with pytest.raises(ValueError):
        s.noise_source = "gpu"
        s.batch_fast_generate(torch.tensor([[128]]).cuda(), auxs, [10])


from conftest import golden_names, load_golden   # noqa: E402


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("g8_")])
def test_seeded_decode_matches_the_reference_fixtures(gpu_ok, name):
    """fixtures g8_*: the REFERENCE's batch_fast_generate with a non-zero `audio` seed, every applicable variant."""
    cfg, d = load_golden(name)
    sd = synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"]))
    net = HipNet.from_state_dict(cfg, sd, "cuda:0")
    n_samples = [int(n) for n in d["n_samples"]]
    aux = torch.from_numpy(d["aux"])
    big = cfg.H == 64
    soft = cfg.kind == "softmax"
    n_steps = max(n_samples) // (1 if soft else cfg.seg)
    noise = torch.from_numpy(d["q"] if soft else d["noise"]).permute(1, 0, 2).contiguous()
    seed = torch.from_numpy(d["seed"]).reshape(len(n_samples), -1)
    for v in ((2, 1) if big else (1, 3)):
        out, _ = net.decode(aux, n_steps, noise, variant=v, seed=seed[:, 0] if soft else seed)
        for b, n in enumerate(n_samples):
            got = out[b, :n].cpu().numpy()
            if soft:
                assert np.array_equal(got, d[f"samples_{b}"]), (name, v, b)
            else:
                assert np.abs(got - d[f"samples_{b}"]).max() <= 1e-5, (name, v, b)
