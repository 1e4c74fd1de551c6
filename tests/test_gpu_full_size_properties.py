"""GPU: the decode hot path at BASELINE.json's full size (cfg2: one utterance of Tf = 600 frames = 66 000 sampling
steps, and cfg5's 64-utterance share), checked through properties that do not need a 66 000-step CPU reference:

  * sampling law - every sample is clamp(mu - sigmoid(s) * sign(e) * log1p(-2|e|), -1, 1) of the heads the kernel
    reports for that step and the deviate it was given (cswnv_shift1.py:386-391);
  * decode == teacher-forced stack - feeding the generated waveform to the parallel forward (csrc/swn_stack.hip, an
    independent set of kernels) reproduces the per-step heads of the autoregressive kernel to 2e-5 (past the warm-up,
    where the reference's own generate and forward differ by their padding): errors cannot
    hide behind free-running divergence because both sides condition on the same samples;
  * prefix - the first m samples of an n-step decode are bit-identical to an m-step decode (causality, no dependence
    on the launch length);
  * batch independence - an utterance decoded alone and inside a 64-utterance batch gives bit-identical samples.
The fixtures (tests/golden) pin the same kernels against the reference itself at sizes it finishes in seconds."""
import numpy as np
import pytest
import torch

from shallow_wavenet_amd import config as C
from shallow_wavenet_amd import noise as NZ
from shallow_wavenet_amd.runtime import HipNet
from shallow_wavenet_amd.synth import synth_aux, synth_state_dict

pytestmark = pytest.mark.gpu

TF = 600


@pytest.fixture(scope="module")
def full():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    cfg = C.bl6_laplace(seg=1, lpc=0)
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True), "cuda:0")
    n = TF * cfg.U
    aux = torch.from_numpy(synth_aux(cfg, 1, TF, seed=1)).cuda()
    noise = NZ.laplace_uniform(cfg, n, 1, generator=torch.Generator().manual_seed(1)).cuda()
    out, heads = net.decode(aux, n, noise, want_heads=True)
    return cfg, net, aux, noise, out, heads, n


def test_samples_follow_the_sampling_law(gpu_ok, full):
    cfg, net, aux, noise, out, heads, n = full
    h = heads[0].double().cpu().numpy()
    e = noise[0, :, 0].double().cpu().numpy()
    mu, b = h[:, 0], 1.0 / (1.0 + np.exp(-h[:, 1]))
    want = np.clip(mu - b * np.sign(e) * np.log1p(-2.0 * np.abs(e)), -1.0, 1.0)
    got = out[0].double().cpu().numpy()
    assert got.shape == (n,) and np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-6
    assert got.std() > 1e-3, "degenerate waveform"


def test_decode_heads_equal_the_teacher_forced_stack(gpu_ok, full):
    cfg, net, aux, noise, out, heads, n = full
    # forward(): audio (B, 1, T - seg) = the generated samples; raw[:, :, t] are the head parameters of the sample that
    # follows audio[t], i.e. of decode step t + 1.  The first positions differ by construction: the decode prologue runs
    # on replicate-padded conditioning (cswnv_shift1.py:297-334), the stack zero-pads its causal history - past two
    # receptive fields both see the same samples and frames only.
    audio = out[:, None, :n - 1].contiguous()
    raw = net.forward(aux, audio)
    raw = raw[0] if isinstance(raw, tuple) else raw
    Tp = raw.shape[2]
    assert Tp == n - 1
    lo = 2 * cfg.receptive_field
    d = (raw[0].transpose(0, 1)[lo:] - heads[0, lo + 1:Tp + 1]).abs().max().item()
    assert d <= 2e-5, d


def test_prefix_of_a_longer_decode_is_identical(gpu_ok, full):
    cfg, net, aux, noise, out, heads, n = full
    m = 12345
    short, _ = net.decode(aux, m, noise[:, :m].contiguous())
    assert torch.equal(short[0], out[0, :m])


def test_utterance_is_independent_of_its_batch(gpu_ok, full):
    cfg, net, aux, noise, out, heads, n = full
    B, frames = 64, 60
    steps = frames * cfg.U
    auxb = torch.from_numpy(synth_aux(cfg, B, frames, seed=5)).cuda()
    nb = NZ.laplace_uniform(cfg, steps, B, generator=torch.Generator().manual_seed(5)).cuda()
    allb, _ = net.decode(auxb, steps, nb)
    for b in (0, 17, 63):
        one, _ = net.decode(auxb[b:b + 1].contiguous(), steps, nb[b:b + 1].contiguous())
        assert torch.equal(one[0], allb[b]), b


def test_softmax_full_size_sampling_and_stack(gpu_ok):
    """cfg1 (softmax mu-law 256, 16 kHz, Tf = 600 -> 48 000 steps): every index is the argmax of softmax(logits) / q of
    the logits the kernel reports for that step (dswnv.py:361-369 == argmax(p / q), q ~ Exp(1)); and the teacher-forced
    stack fed with the generated indices reproduces those logits (2e-4, as in the fixture tests) past the warm-up."""
    cfg = C.bl6_softmax()
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True), "cuda:0")
    n = TF * cfg.U
    aux = torch.from_numpy(synth_aux(cfg, 1, TF, seed=2)).cuda()
    q = NZ.softmax_exponential(cfg, n, 1, generator=torch.Generator().manual_seed(2)).cuda()
    idx, logits = net.decode(aux, n, q, want_heads=True)
    assert idx.shape == (1, n) and int(idx.min()) >= 0 and int(idx.max()) < cfg.n_quantize
    lg = logits[0].double()
    score = torch.log_softmax(lg, dim=1) - torch.log(q[0].double())          # argmax(p / q), monotone transform
    want = score.argmax(dim=1)
    got = idx[0].long()
    bad = (want != got).nonzero().flatten()
    if bad.numel():                                                           # only exact near-ties may differ
        top2 = score[bad].topk(2, dim=1).values
        assert float((top2[:, 0] - top2[:, 1]).max()) <= 1e-6, (bad.numel(), float((top2[:, 0] - top2[:, 1]).max()))
    assert bad.numel() <= 2
    assert got.unique().numel() > 16, "degenerate index stream"
    raw = net.forward(aux, idx[:, :n - 1].contiguous())
    raw = raw[0] if isinstance(raw, tuple) else raw
    lo = 2 * cfg.receptive_field
    d = (raw[0].transpose(0, 1)[lo:] - logits[0, lo + 1:n]).abs().max().item()
    assert d <= 2e-4, d


def test_cfg3_full_size_lpc_recursion_and_stack(gpu_ok):
    """cfg3 (BL6 Laplace, seg = 5, lpc = 4, Tf = 600 -> 13 200 steps of 5 samples) at its full size:
      * LP recursion law per sub-step (cswnv_shift1.py:368-385): with a = heads[2seg:] used FLIPPED against the last
        lpc generated samples (zeros before the start), s_j = clamp(a . buf + mu_j - sigmoid(.)_j * sign(e) log1p(-2|e|));
      * the teacher-forced stack fed with the generated waveform reproduces the heads of step i at its position
        (i-1)*seg past two receptive fields;
      * a prefix of the decode is bit-identical to the shorter decode."""
    cfg = C.bl6_laplace(seg=5, lpc=4)
    seg, lpc = cfg.seg, cfg.lpc
    net = HipNet.from_state_dict(cfg, synth_state_dict(cfg, seed=1, flavor="trained", identity_scale_in=True), "cuda:0")
    n_steps = TF * cfg.U // seg
    n = n_steps * seg
    aux = torch.from_numpy(synth_aux(cfg, 1, TF, seed=3)).cuda()
    noise = NZ.laplace_uniform(cfg, n_steps, 1, generator=torch.Generator().manual_seed(3)).cuda()
    out, heads = net.decode(aux, n_steps, noise, want_heads=True)
    assert out.shape == (1, n) and heads.shape == (1, n_steps, 2 * seg + lpc)
    got = out[0].double().cpu().numpy()
    h = heads[0].double().cpu().numpy()
    e = noise[0].double().cpu().numpy().reshape(-1)                              # (step, j) order
    mu = h[:, :seg].reshape(-1)
    b = (1.0 / (1.0 + np.exp(-h[:, seg:2 * seg]))).reshape(-1)
    a = np.repeat(h[:, 2 * seg:], seg, axis=0)                                   # (n, lpc), the step's coefficients
    hist = np.concatenate([np.zeros(lpc), got])                                  # hist[p + k] = x[p - lpc + k]
    win = np.lib.stride_tricks.sliding_window_view(hist, lpc)[:n]                # win[p, k] = x[p - lpc + k]
    pred = (a[:, ::-1] * win).sum(1)
    want = np.clip(pred + mu - b * np.sign(e) * np.log1p(-2.0 * np.abs(e)), -1.0, 1.0)
    assert np.isfinite(got).all() and got.std() > 1e-3
    assert np.abs(got - want).max() <= 5e-6, np.abs(got - want).max()
    audio = out[:, None, : n - seg].contiguous()
    raw = net.forward(aux, audio)
    raw = raw[0] if isinstance(raw, tuple) else raw
    assert raw.shape[2] == n - 2 * seg + 1
    i0 = 2 * cfg.receptive_field // seg + 2
    steps = torch.arange(i0, n_steps - 1, device=raw.device)
    d = (raw[0].transpose(0, 1)[(steps - 1) * seg] - heads[0, steps]).abs().max().item()
    assert d <= 2e-5, d
    m = 2345
    short, _ = net.decode(aux, m, noise[:, :m].contiguous())
    assert torch.equal(short[0], out[0, : m * seg])
