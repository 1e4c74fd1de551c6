"""CPU: pin the oracle (oracle/cpu_ref.py) against every golden fixture that was produced by
importing the reference's own src/nets modules (oracle/make_golden.py).

Tolerances: free-running Laplace samples <= 1e-5 abs (north_star), head outputs <= 1e-5,
softmax indices bit-exact.
"""
import ast

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import cpu_ref
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.synth import synth_state_dict

TOL = 1e-5
MAX_STEPS_BIG = 300       # REF6 fixtures: check the first steps only, keeps the CPU suite short


def _params(cfg, d):
    sd = synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"]))
    return cpu_ref.as_params(sd)


LAP = [n for n in golden_names() if "_lap_" in n and not n.startswith(("g5_", "g6_", "g9_"))]
SMX = [n for n in golden_names() if "softmax" in n and not n.startswith(("g5_", "g6_", "g9_"))]


def test_fixture_inventory():
    names = golden_names()
    assert len([n for n in names if n.startswith("g0_")]) >= 13
    assert len([n for n in names if n.startswith("g1_")]) >= 7
    assert "g3_numerics" in names


@pytest.mark.parametrize("name", LAP)
def test_laplace_generate_matches_reference(name):
    cfg, d = load_golden(name)
    P = _params(cfg, d)
    n_samples = [int(n) for n in d["n_samples"]]
    big = name.startswith("g2_")
    if big:
        n_samples = [min(n, MAX_STEPS_BIG * cfg.seg) for n in n_samples]
    res, heads = cpu_ref.laplace_generate(cfg, P, torch.from_numpy(d["aux"]), n_samples, d["noise"],
                                          return_heads=True)
    n_steps = heads.shape[0]
    assert np.abs(heads - d["heads"][:n_steps]).max() <= TOL
    for b, n in enumerate(n_samples):
        ref = d[f"samples_{b}"][:n]
        assert res[b].shape == ref.shape
        assert np.abs(res[b] - ref).max() <= TOL, name


@pytest.mark.parametrize("name", [n for n in LAP if not n.startswith("g2_")])
def test_laplace_noise_order(name):
    cfg, d = load_golden(name)
    g = torch.Generator().manual_seed(int(d["noise_seed"]))
    noise = cpu_ref.laplace_noise(cfg, d["noise"].shape[0], d["noise"].shape[1], generator=g)
    assert np.array_equal(noise, d["noise"])


@pytest.mark.parametrize("name", [n for n in LAP if "fwd_0" in load_golden(n)[1]])
def test_laplace_forward_matches_reference(name):
    cfg, d = load_golden(name)
    P = _params(cfg, d)
    res = cpu_ref.laplace_forward(cfg, P, torch.from_numpy(d["aux"]), torch.from_numpy(d["fwd_audio"]))
    n_ret = 4 if cfg.lpc > 0 else 3
    assert len(res) == n_ret
    for i, r in enumerate(res):
        ref = d[f"fwd_{i}"]
        assert tuple(r.shape) == ref.shape
        assert np.abs(r.numpy() - ref).max() <= TOL
    resc = cpu_ref.laplace_forward(cfg, P, torch.from_numpy(d["aux"]), torch.from_numpy(d["fwd_audio"]),
                                   clip=True)
    assert len(resc) == int(d["fwd_clip_n"])


@pytest.mark.parametrize("name", [n for n in LAP if "loss" in load_golden(n)[1]])
def test_laplace_backward_matches_reference(name):
    cfg, d = load_golden(name)
    P = _params(cfg, d)
    for v in P.values():
        v.requires_grad_(True)
    res = cpu_ref.laplace_forward(cfg, P, torch.from_numpy(d["aux"]), torch.from_numpy(d["fwd_audio"]))
    loss = cpu_ref.laplace_nll(res[0], res[1], torch.from_numpy(d["loss_target"]), log_b=res[2])
    if cfg.lpc > 0:
        loss = loss + 0.1 * res[3].pow(2).mean()
    assert abs(loss.item() - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    loss.backward()
    for k, v in P.items():
        g = v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape), np.float32)
        dig = d[f"gdig_{k}"]
        scale = max(1e-3, dig[1])
        assert abs(g.astype(np.float64).sum() - dig[0]) <= 2e-4 * scale, k
        assert abs(np.abs(g.astype(np.float64)).sum() - dig[1]) <= 2e-4 * scale, k
        if f"grad_{k}" in d:
            assert np.abs(g - d[f"grad_{k}"]).max() <= 1e-5 + 1e-4 * np.abs(d[f"grad_{k}"]).max(), k


@pytest.mark.parametrize("name", [n for n in LAP if "solo_samples" in load_golden(n)[1]])
def test_batch_padding_quirk_pinned(name):
    """G4: a short utterance decoded inside a longer zero-padded batch differs from the same
    utterance decoded alone only where the +-4-frame conditioning context sees the padding."""
    cfg, d = load_golden(name)
    b = int(d["solo_index"])
    in_batch, solo = d[f"samples_{b}"], d["solo_samples"]
    assert in_batch.shape == solo.shape
    frames = int(d["frames"][b])
    ctx = (cfg.aux_kernel_size ** cfg.aux_dilation_size) // 2
    clean = max((frames - ctx) * cfg.U - cfg.seg, 0)
    if clean > 0:
        assert np.abs(in_batch[:clean] - solo[:clean]).max() <= TOL
    if frames < int(d["frames"].max()):
        assert np.abs(in_batch[clean:] - solo[clean:]).max() > 1e-4   # the quirk is real


@pytest.mark.parametrize("name", SMX)
def test_softmax_generate_matches_reference(name):
    cfg, d = load_golden(name)
    P = _params(cfg, d)
    n_samples = [int(n) for n in d["n_samples"]]
    if name.startswith("g2_"):
        n_samples = [min(n, MAX_STEPS_BIG) for n in n_samples]
    n_steps = max(n_samples)
    if "q" in d:
        q = d["q"]
    else:
        g = torch.Generator().manual_seed(int(d["noise_seed"]))
        q = cpu_ref.softmax_noise(cfg, int(d["n_samples"].max()), len(n_samples), generator=g)
    res, heads, margins = cpu_ref.softmax_generate(cfg, P, torch.from_numpy(d["aux"]), n_samples, q,
                                                   return_heads=True)
    st = int(d["head_stride"])
    ref_heads = d["heads"]
    assert np.abs(heads[:n_steps:st] - ref_heads[: len(heads[:n_steps:st])]).max() <= 2e-5
    for b, n in enumerate(n_samples):
        assert np.array_equal(res[b], d[f"samples_{b}"][:n]), name


@pytest.mark.parametrize("name", [n for n in SMX if "fwd_audio_idx" in load_golden(n)[1]])
def test_softmax_forward_matches_reference(name):
    cfg, d = load_golden(name)
    P = _params(cfg, d)
    idx = torch.from_numpy(d["fwd_audio_idx"])
    oh = cpu_ref.one_hot(idx, cfg.n_quantize).transpose(1, 2)
    logits = cpu_ref.softmax_forward(cfg, P, oh, torch.from_numpy(d["aux"])).detach().numpy()
    assert np.abs(logits[:, :64] - d["fwd_logits_head"]).max() <= 2e-5
    assert np.abs(logits[:, -64:] - d["fwd_logits_tail"]).max() <= 2e-5
    dig = d["fwd_logits_dig"]
    assert abs(logits.astype(np.float64).sum() - dig[0]) <= 1e-5 * dig[1]


def test_numerics_tables():
    _, d = load_golden("g3_numerics")
    idx = np.arange(256)
    assert np.array_equal(cpu_ref.decode_mu_law(idx, 256), d["mulaw_decode_256"])
    assert np.array_equal(cpu_ref.encode_mu_law(d["mulaw_sweep"], 256), d["mulaw_encode_sweep"])
    assert abs(cpu_ref.decode_mu_law(0) - (-1.0221)) < 1e-3 and abs(cpu_ref.decode_mu_law(255) - 0.9784) < 1e-3
    t = cpu_ref.laplace_transform(torch.from_numpy(d["lap_eps"])).numpy()
    assert np.array_equal(t, d["lap_t"])
    oh = cpu_ref.one_hot(torch.tensor([[0, 255, 256, 511, -1, 128]]), 256)
    assert np.array_equal(oh.argmax(-1).numpy(), d["onehot_argmax"])


def test_geometry_matches_reference_modules():
    _, d = load_golden("g3_numerics")
    table = {"bl6_laplace": C.bl6_laplace(), "bl6_laplace_s5l4": C.bl6_laplace(5, 4),
             "bl6_softmax": C.bl6_softmax(), "ref6_laplace": C.ref6_laplace(),
             "ref6_laplace_s5": C.ref6_laplace(5, 4), "ref6_softmax": C.ref6_softmax(),
             "tiny_laplace": C.tiny(), "tiny_softmax": C.tiny("softmax", wav_conv_flag=False)}
    for row in d["geometry"]:
        g = ast.literal_eval(str(row))
        cfg = table[g["name"]]
        assert cfg.receptive_field == g["rf"]
        assert cfg.paddings == g["padding"]
        assert cfg.n_params() == g["n_params"]
        assert [(k, tuple(s)) for k, s in cfg.param_shapes()] == [(k, tuple(s)) for k, s in g["keys"]]
    assert C.ref6_laplace().receptive_field == 690      # run.sh:179 comment says 691
    assert C.bl6_laplace().receptive_field == 64


DROP = [n for n in golden_names() if n.startswith("g5_drop") or n.startswith("g9_drop")]


@pytest.mark.parametrize("name", DROP)
def test_dropout_masks_and_oracle_reproduce_the_reference_training_forward(name):
    """model.train(), forward(do=True): the masks re-drawn on the host in the reference's order
    (noise.dropout_masks) + the oracle's masked stack give the reference's own outputs."""
    from shallow_wavenet_amd import noise as swn_noise
    cfg, d = load_golden(name)
    assert len(DROP) >= 4
    P = _params(cfg, d)
    B, Tf = d["aux"].shape[0], d["aux"].shape[2]
    torch.manual_seed(int(d["drop_seed"]))
    do = bool(int(d["do"])) if "do" in d else True
    drop = swn_noise.dropout_masks(cfg, B, Tf, float(d["drop_p"]), draw_x=do)
    keep = 1.0 - float(d["drop_p"])
    vals = set(np.unique(drop[0].numpy()).tolist())
    assert vals <= ({0.0, np.float32(1.0 / keep).item()} if do else {1.0})
    assert [m is not None for m in drop[1]] == [l in swn_noise.dropped_layers(cfg) and l + 1 < cfg.L for l in range(cfg.L)]
    if cfg.kind == "laplace":
        raw, _ = cpu_ref.laplace_stack(cfg, P, torch.from_numpy(d["aux"]), torch.from_numpy(d["fwd_audio"]), drop=drop)
        mu = raw.transpose(1, 2)[:, :, :cfg.seg].reshape(d["fwd_0"].shape)
        assert np.abs(mu.numpy() - d["fwd_0"]).max() <= TOL
    else:
        raw, _ = cpu_ref.softmax_stack(cfg, P, torch.from_numpy(d["fwd_audio_idx"]), torch.from_numpy(d["aux"]), drop=drop)
        lg = raw.transpose(1, 2).numpy()
        assert np.abs(lg[:, :64] - d["fwd_logits_head"]).max() <= 2e-4
        assert np.abs(lg[:, -64:] - d["fwd_logits_tail"]).max() <= 2e-4


# ---- G7: teacher-forced forward + backward AT THE run.sh GEOMETRY (H=192/256, K=7), produced by the reference
G7 = [n for n in golden_names() if n.startswith("g7_")]


def grad_sample_index(size, n=2048):
    """same subset rule as oracle/make_golden.py::grad_sample_index"""
    return np.unique(np.linspace(0, size - 1, min(n, size)).astype(np.int64))


def check_grads_against_fixture(name, grads, d, atol=2e-5, rtol=2e-4):
    """grads: {state_dict key: ndarray}.  Elementwise on what the fixture stores (whole tensor or the fixed subset),
    digests on the rest."""
    for k, g in grads.items():
        g = np.asarray(g, dtype=np.float64)
        dig = d[f"gdig_{k}"]
        scale = max(1e-3, dig[1])
        assert abs(g.sum() - dig[0]) <= 5e-4 * scale, (name, k, g.sum(), dig[0])
        assert abs(np.abs(g).sum() - dig[1]) <= 5e-4 * scale, (name, k)
        if f"grad_{k}" in d:
            ref, got = d[f"grad_{k}"], g
        else:
            ref, got = d[f"gsamp_{k}"], g.ravel()[grad_sample_index(g.size)]
        assert np.abs(got - ref).max() <= atol + rtol * np.abs(ref).max(), (name, k, np.abs(got - ref).max())


@pytest.mark.parametrize("name", G7)
def test_oracle_teacher_forced_ref6_forward_and_gradients(name):
    """the oracle's stack, differentiated by torch autograd on the CPU, against the reference's own outputs, loss
    and loss.backward() at REF6 size (B=2 ragged, 990 positions, rf=690)."""
    cfg, d = load_golden(name)
    P = {k: v.clone().requires_grad_(True) for k, v in _params(cfg, d).items()}
    aux = torch.from_numpy(d["aux"])
    if cfg.kind == "laplace":
        res = cpu_ref.laplace_forward(cfg, P, aux, torch.from_numpy(d["fwd_audio"]))
        for i, r in enumerate(res):
            assert np.abs(r.detach().numpy() - d[f"fwd_{i}"]).max() <= TOL, (name, i)
        assert len(cpu_ref.laplace_forward(cfg, P, aux, torch.from_numpy(d["fwd_audio"]), clip=True)) == int(d["fwd_clip_n"])
        loss = cpu_ref.laplace_nll(res[0], res[1], torch.from_numpy(d["loss_target"]), log_b=res[2])
        if cfg.lpc > 0:
            loss = loss + 0.1 * res[3].pow(2).mean()
    else:
        oh = cpu_ref.one_hot(torch.from_numpy(d["fwd_audio_idx"]), cfg.n_quantize).transpose(1, 2)
        logits = cpu_ref.softmax_forward(cfg, P, oh, aux)
        ln = logits.detach().numpy()
        assert np.abs(ln[:, :64] - d["fwd_logits_head"]).max() <= 2e-5
        assert np.abs(ln[:, -64:] - d["fwd_logits_tail"]).max() <= 2e-5
        assert np.abs(ln[:, ::16] - d["fwd_logits_s16"]).max() <= 2e-5
        loss = torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), torch.from_numpy(d["loss_target"]).reshape(-1))
    assert abs(loss.item() - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    loss.backward()
    grads = {k: (P[k].grad.numpy() if P[k].grad is not None else np.zeros(tuple(P[k].shape), np.float32)) for k in P}
    check_grads_against_fixture(name, grads, d)


G9 = [n for n in golden_names() if n.startswith("g9_drop")]


def drop_masks_of(cfg, d):
    """the masks the reference drew for a g5 / g9 fixture (host stream, reference order)."""
    from shallow_wavenet_amd import noise as swn_noise
    torch.manual_seed(int(d["drop_seed"]))
    do = bool(int(d["do"])) if "do" in d else True
    return swn_noise.dropout_masks(cfg, d["aux"].shape[0], d["aux"].shape[2], float(d["drop_p"]), draw_x=do)


def oracle_dropout_loss(cfg, P, d):
    """loss of a g5 / g9 fixture through the oracle with the reference's masks -> (loss, outputs)."""
    drop = drop_masks_of(cfg, d)
    aux = torch.from_numpy(d["aux"])
    if cfg.kind == "laplace":
        res = cpu_ref.laplace_forward(cfg, P, aux, torch.from_numpy(d["fwd_audio"]), drop=drop)
        loss = cpu_ref.laplace_nll(res[0], res[1], torch.from_numpy(d["loss_target"]), log_b=res[2])
        if cfg.lpc > 0:
            loss = loss + 0.1 * res[3].pow(2).mean()
        return loss, res
    raw, _ = cpu_ref.softmax_stack(cfg, P, torch.from_numpy(d["fwd_audio_idx"]), aux, drop=drop)
    logits = raw.transpose(1, 2)
    return torch.nn.CrossEntropyLoss()(logits.reshape(-1, cfg.n_quantize), torch.from_numpy(d["loss_target"]).reshape(-1)), logits


@pytest.mark.parametrize("name", G9)
def test_oracle_dropout_mode_at_the_trained_geometries(name):
    """G9: model.train(), forward(do=True), do_prob = 0.5 (run.sh:198) through the REFERENCE at REF6 Laplace / BL6 Laplace /
    REF6 softmax; the oracle with the re-drawn masks, differentiated by autograd, gives its outputs, loss and gradients."""
    cfg, d = load_golden(name)
    assert len(G9) == 3 and float(d["drop_p"]) == 0.5
    P = {k: v.clone().requires_grad_(True) for k, v in _params(cfg, d).items()}
    loss, res = oracle_dropout_loss(cfg, P, d)
    if cfg.kind == "laplace":
        for i, r in enumerate(res):
            assert np.abs(r.detach().numpy() - d[f"fwd_{i}"]).max() <= TOL, (name, i)
    else:
        ln = res.detach().numpy()
        assert np.abs(ln[:, ::16] - d["fwd_logits_s16"]).max() <= 2e-5
    assert abs(loss.item() - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    loss.backward()
    grads = {k: (P[k].grad.numpy() if P[k].grad is not None else np.zeros(tuple(P[k].shape), np.float32)) for k in P}
    check_grads_against_fixture(name, grads, d)


def test_philox_restatement_known_answers():
    """Random123's published Philox4x32-10 known-answer vectors (kat_vectors: counter, key -> output)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = cpu_ref.philox4x32_10(np.array(ctr, dtype=np.uint32), key)
        assert tuple(int(x) for x in got) == want
    e = cpu_ref.device_noise("laplace", 0x1234567890ABCDEF, 3, 2, 5, 5)
    assert e.shape == (2, 5, 5) and e.dtype == np.float32 and e.min() >= -0.4999 and e.max() < 0.5
    q = cpu_ref.device_noise("softmax", 7, 0, 1, 3, 256)
    assert q.shape == (1, 3, 256) and q.min() > 0 and abs(q.mean() - 1.0) < 0.2


G8 = [n for n in golden_names() if n.startswith("g8_")]


@pytest.mark.parametrize("name", G8)
def test_oracle_generate_with_a_nonzero_seed_matches_the_reference(name):
    """batch_fast_generate(audio != 0) of the reference (cswnv_shift1.py:300-334 / dswnv.py:305-336)."""
    cfg, d = load_golden(name)
    P = _params(cfg, d)
    n_samples = [int(n) for n in d["n_samples"]]
    aux = torch.from_numpy(d["aux"])
    if cfg.kind == "laplace":
        res = cpu_ref.laplace_generate(cfg, P, aux, n_samples, d["noise"], seed=d["seed"])
        zero = cpu_ref.laplace_generate(cfg, P, aux, n_samples, d["noise"])
        for b, n in enumerate(n_samples):
            assert np.abs(res[b] - d[f"samples_{b}"]).max() <= TOL, (name, b)
        assert np.abs(zero[0] - d["samples_0"]).max() > 1e-4           # the seed is not ignored
    else:
        res = cpu_ref.softmax_generate(cfg, P, aux, n_samples, d["q"], seed=d["seed"])
        for b, n in enumerate(n_samples):
            assert np.array_equal(res[b], d[f"samples_{b}"]), (name, b)
