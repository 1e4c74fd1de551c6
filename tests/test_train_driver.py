"""CPU: host logic of the stage-4 training driver (shallow_wavenet_amd/train_driver.py) - chunk plan, length
validation, generator protocol, optimizer parameter list, checkpoint dictionary, CLI surface.  The numerical
side (one chunk through the HIP forward/backward) is tests/test_gpu_train_step.py."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from shallow_wavenet_amd import train_driver as T
from shallow_wavenet_amd.nets import cswnv_shift1 as mc


def test_chunk_plan_matches_fixture_and_tiles_the_utterance():
    for name in [n for n in golden_names() if n.startswith("g6_trainstep") and "softmax" not in n]:
        cfg, d = load_golden(name)
        plan = T.chunk_plan(d["h"].shape[0], cfg.receptive_field, int(d["batch_size"]), cfg.seg, cfg.U)
        assert np.array_equal(np.array(plan), d["plan"])
    # reference defaults: rf=690 (REF6), batch 8800, U=110, seg=5 (train_cswnv...py:96-145)
    rf, bs, seg, U = 690, 8800, 5, 110
    plan = T.chunk_plan(400, rf, bs, seg, U)
    chunk = rf + bs + seg
    assert plan[0] == (chunk // U, chunk // U * U, 0, 0)
    delta = T.effective_batch_size(rf, bs, seg, U) // U
    for i, (h_bs, x_bs, h_ss, x_ss) in enumerate(plan):
        assert h_ss == i * delta and x_ss == h_ss * U
        assert (h_bs, x_bs) == ((chunk // U, chunk // U * U) if i + 1 < len(plan) else (-1, -1))
    # predicted regions [x_ss + rf + seg, end) of consecutive chunks leave no gap
    ends = [x_ss + x_bs if x_bs > 0 else 400 * U for (_, x_bs, _, x_ss) in plan]
    starts = [x_ss + (rf if x_ss > 0 else 0) + seg for (_, _, _, x_ss) in plan]
    assert all(s <= e for s, e in zip(starts[1:], ends[:-1]))
    assert T.chunk_plan(5, rf, bs, seg, U) == []           # too short for even rf + 2 seg


def test_validate_length_and_fft_sizes():
    x, h = T.validate_length(np.zeros(1000), np.zeros((10, 3)), 110)
    assert len(x) == 990 and len(h) == 9
    x, h = T.validate_length(np.zeros(1100), np.zeros((8, 3)), 110)
    assert len(x) == 880 and len(h) == 8
    assert T.fft_sizes(17)[0] == 128 and T.fft_sizes(17)[-1] == 2048 and len(T.fft_sizes(9)) == 9
    assert T.fft_sizes(4) == [128, 192, 256, 384]


def test_generator_protocol_and_epoch_marker():
    names, feats, loader = T.synthetic_corpus(3, 10, 20, min_frames=30, max_frames=40, seed=1)
    np.random.seed(3)
    gen = T.train_generator(names, feats, 54, "/feat_org_lf0", 200, 1, True, 20, None, loader)
    seen, rec = [], next(gen)
    while rec[2] >= 0:
        x, h, c_idx, utt_idx, wav, h_bs, x_bs, h_ss, x_ss = rec
        assert x.shape[0] == h.shape[0] * 20 and wav == names[utt_idx]
        seen.append(utt_idx)
        rec = next(gen)
    assert sorted(set(seen)) == [0, 1, 2]                  # every utterance once per epoch, then the c_idx = -1 record
    assert next(gen)[2] == 0                               # next epoch starts
    ev = T.train_generator(names, feats, 54, "/feat_org_lf0", 200, 1, False, 20, None, loader)
    assert next(ev)[3] == 0                                # evaluation keeps the list order


def test_optimizer_list_excludes_scale_in_and_checkpoint_keys(tmp_path):
    m = mc.CSWNV(n_aux=10, hid_chn=32, skip_chn=48, dilation_depth=3, dilation_repeat=2, kernel_size=3,
                 upsampling_factor=20, seg=2, lpc=4, aux_conv2d_flag=True, wav_conv_flag=True)
    T.set_scale_in(m, np.zeros(10), np.full(10, 2.0))
    assert torch.allclose(m.scale_in.weight[:, :, 0], torch.eye(10) * 0.5)
    assert not any(p.requires_grad for p in m.scale_in.parameters())
    plist = T.optimizer_parameters(m)
    ids = {id(p) for p in plist}
    assert all(id(p) not in ids for p in m.scale_in.parameters())
    assert len(plist) == len(list(m.parameters())) - 2
    assert plist[0] is next(m.conv_aux.parameters())
    opt = torch.optim.Adam(plist, lr=1e-4)
    T.save_checkpoint(str(tmp_path), m, opt, np.random.get_state(), torch.get_rng_state(), 7)
    ck = torch.load(str(tmp_path / "checkpoint-7.pkl"), weights_only=True)
    assert set(ck) == {"model", "optimizer", "numpy_random_state", "torch_random_state", "iterations"}
    assert list(ck["model"]) == list(m.state_dict()) and ck["iterations"] == 7


def test_cli_flags_are_the_reference_flags():
    """train_cswnv_laplace-stftcmplx_shift1.py:185-249"""
    want = {"waveforms", "waveforms_eval", "feats", "feats_eval", "stats", "expdir", "n_aux", "skip_chn", "seg",
            "dilation_depth", "dilation_repeat", "hid_chn", "kernel_size", "aux_kernel_size", "aux_dilation_size",
            "upsampling_factor", "n_fft_facts", "string_path", "lr", "batch_size", "epoch_count", "do_prob", "lpc",
            "aux_conv2d_flag", "wav_conv_flag", "seed", "resume", "pretrained", "GPU_device", "verbose"}
    have = {a.dest for a in T.build_parser()._actions}
    assert want <= have
    ns = T.build_parser().parse_args(["--expdir", "x", "--aux_conv2d_flag", "true", "--wav_conv_flag", "false"])
    assert ns.aux_conv2d_flag is True and ns.wav_conv_flag is False and ns.lr == 1e-4 and ns.batch_size == 8800
    # not a reference flag: arithmetic of the training contractions, parity mode unless asked otherwise
    assert ns.precision == "fp32"
    assert T.build_parser().parse_args(["--expdir", "x", "--precision", "bf16"]).precision == "bf16"
    with pytest.raises(SystemExit):
        T.build_parser().parse_args(["--expdir", "x", "--precision", "fp8"])


def test_softmax_driver_chunk_plan_and_flags():
    from shallow_wavenet_amd import train_softmax_driver as S
    # train_dswnv_softmax.py:96-146 with the recipe's defaults (rf of the 3x3, K=6 stack = 520+..., batch 1100, U=110)
    rf, bs, U = 265, 1100, 110
    plan = S.chunk_plan(60, rf, bs, U)
    eff = bs - (rf + bs + 1) % U
    h_bs = (rf + eff + 1) // U
    assert plan[0] == (h_bs, h_bs * U, 0, 0) and plan[-1][:2] == (-1, -1)
    assert all(p[2] == i * (eff // U) for i, p in enumerate(plan))
    assert S.chunk_plan(2, rf, bs, U) == []                     # 220 samples <= rf + 1: nothing to train on
    have = {a.dest for a in S.build_parser()._actions}
    assert {"n_quantize", "audio_in", "wav_conv_flag", "do_prob", "batch_size", "epoch_count", "stats", "resume"} <= have
    assert "precision" in have
    x = torch.arange(700)
    h = torch.zeros(40, 3)
    bh, bx, trg = S.slice_chunk(x, h, 17, 340, 14, 280)
    assert bh.shape == (1, 3, 17) and bx.shape == (1, 339) and trg.shape == (339,)
    assert int(bx[0, 0]) == 280 and int(trg[0]) == 281 and int(trg[-1]) == 280 + 339
