"""GPU: one stage-4 training chunk (train_cswnv_laplace-stftcmplx_shift1.py:700-874) through the drop-in module:
dropout masks + LP mean by unfold + Laplace NLL + reparameterised-sample complex-STFT L1, backward through the HIP
kernels, against fixtures g6_trainstep_* computed with the REFERENCE's CSWNV / LaplaceLoss / LSDloss on the CPU under
the same loss assembly (shallow_wavenet_amd/train_driver.py; the reference script cannot be imported here, so the
assembly itself is pinned only through these module-level fixtures).  Losses to 2e-5 relative, gradients to the
tolerances of test_gpu_backward_parity.py; then an Adam step on the reference's parameter list must run."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from shallow_wavenet_amd import train_driver as T
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.synth import synth_state_dict
from test_gpu_backward_parity import _check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("g6_trainstep") and "softmax" not in n])
def test_training_chunk_matches_reference_modules(gpu_ok, name):
    cfg, d = load_golden(name)
    m = mc.CSWNV(**cfg.ctor_kwargs(), do_prob=float(d["drop_p"]))
    m.dropout_source = "host"            # fixture = the reference's CPU masks
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=int(d["wseed"]), flavor="trained").items()})
    m.cuda().train()
    for p in m.scale_in.parameters():
        p.requires_grad = False
    plan = [tuple(int(v) for v in r) for r in d["plan"]]
    h_bs, x_bs, h_ss, x_ss = plan[int(d["chunk_index"])]
    bh, bx, trg, xp, flen = T.slice_chunk(m, torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["h"]).cuda(), h_bs, x_bs, h_ss, x_ss)
    assert flen == int(d["feat_len"])
    fft = T.fft_sizes(int(d["n_fft_facts"]))
    win = [torch.hann_window(n).cuda() for n in fft]
    torch.manual_seed(int(d["step_seed"]))
    loss, l_lap, l_lsd, l_err = T.batch_loss(m, mc.LaplaceLoss(), mc.LSDloss(), bh, bx, trg, xp, flen, h_ss, fft, win, do=True)
    rel = lambda a, b: abs(a - b) <= 2e-5 * max(1.0, abs(b))
    assert rel(l_lap.item(), float(d["loss_laplace"])), (l_lap.item(), float(d["loss_laplace"]))
    assert rel(l_err.item(), float(d["loss_err"]))
    assert rel(loss.item(), float(d["loss"])), (loss.item(), float(d["loss"]))
    if not np.isnan(float(d["loss_lsd"])):
        assert abs(l_lsd.item() - float(d["loss_lsd"])) <= 1e-3 * max(1.0, abs(float(d["loss_lsd"])))
    opt = torch.optim.Adam(T.optimizer_parameters(m), lr=1e-4)
    opt.zero_grad()
    loss.backward()
    for k, p in m.named_parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)          # frozen scale_in: the fixture stores zeros
    _check(name, m, d)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    opt.step()
    moved = [k for k, v in m.state_dict().items() if not torch.equal(v, before[k])]
    assert "scale_in.weight" not in moved and "out_2.weight" in moved and "conv_aux.conv.0.weight" in moved


def test_softmax_training_chunk_matches_reference_module(gpu_ok):
    """stage 7 (train_dswnv_softmax.py:549-575): mu-law classes in, dropout, cross entropy past the receptive field."""
    from shallow_wavenet_amd import train_softmax_driver as S
    from shallow_wavenet_amd.nets import dswnv as md
    name = "g6_trainstep_tiny_softmax"
    cfg, d = load_golden(name)
    m = md.DSWNV(**cfg.ctor_kwargs(), do_prob=float(d["drop_p"]))
    m.dropout_source = "host"            # fixture = the reference's CPU masks
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=int(d["wseed"]), flavor="xavier").items()})
    m.cuda().train()
    for p in m.scale_in.parameters():
        p.requires_grad = False
    plan = [tuple(int(v) for v in r) for r in d["plan"]]
    assert plan == S.chunk_plan(d["h"].shape[0], m.receptive_field, int(d["batch_size"]), cfg.U)
    h_bs, x_bs, h_ss, x_ss = plan[int(d["chunk_index"])]
    bh, bx, trg = S.slice_chunk(torch.from_numpy(d["xc"]).cuda(), torch.from_numpy(d["h"]).cuda(), h_bs, x_bs, h_ss, x_ss)
    torch.manual_seed(int(d["step_seed"]))
    loss = S.batch_loss(m, torch.nn.CrossEntropyLoss(), bh, bx, trg, h_ss, do=True)
    assert abs(loss.item() - float(d["loss"])) <= 2e-5 * max(1.0, abs(float(d["loss"])))
    opt = torch.optim.Adam(S.optimizer_parameters(m), lr=1e-4)
    opt.zero_grad()
    loss.backward()
    for k, p in m.named_parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    _check(name, m, d)
    opt.step()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_stage4_driver_runs_a_few_chunks_on_synthetic_data(gpu_ok, tmp_path, caplog, precision):
    """the stage-4 driver end to end (chunk generator, dropout masks, HIP forward/backward behind autograd, NLL + STFT
    losses, Adam, log lines) on generated utterances, in the parity mode and in the mixed-precision mode."""
    from shallow_wavenet_amd.runtime import train_precision
    import logging
    exp = tmp_path / precision
    caplog.set_level(logging.INFO)
    try:
        rc = T.main(["--expdir", str(exp), "--synthetic", "3", "--max_iters", "4", "--n_aux", "10", "--hid_chn", "32",
                     "--skip_chn", "48", "--dilation_depth", "3", "--dilation_repeat", "2", "--kernel_size", "3",
                     "--upsampling_factor", "20", "--seg", "1", "--lpc", "0", "--batch_size", "600", "--n_fft_facts", "5",
                     "--do_prob", "0.5", "--wav_conv_flag", "true", "--epoch_count", "1", "--verbose", "1",
                     "--precision", precision])
    finally:
        from shallow_wavenet_amd.runtime import current_precision
        assert current_precision() == 0    # the driver scopes --precision to its own call
    assert rc == 0
    assert (exp / "model.conf").exists()
    text = caplog.text.lower()              # the driver's per-chunk loss lines (pytest owns the root logger, no file)
    assert "iteration" in text and "nan" not in text


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_stage7_softmax_driver_runs_a_few_chunks_on_synthetic_data(gpu_ok, tmp_path, caplog, precision):
    """the softmax (DSWNV) driver end to end on generated utterances: chunk plan, mu-law classes, one-hot input, HIP
    forward/backward, cross entropy past the receptive field, Adam - both arithmetic modes."""
    import logging
    from shallow_wavenet_amd import train_softmax_driver as S
    from shallow_wavenet_amd.runtime import train_precision
    exp = tmp_path / precision
    caplog.set_level(logging.INFO)
    try:
        rc = S.main(["--expdir", str(exp), "--synthetic", "3", "--max_iters", "4", "--n_aux", "10", "--hid_chn", "32",
                     "--skip_chn", "48", "--dilation_depth", "3", "--dilation_repeat", "2", "--kernel_size", "3",
                     "--upsampling_factor", "20", "--batch_size", "400", "--do_prob", "0.5", "--epoch_count", "1",
                     "--verbose", "1", "--precision", precision])
    finally:
        from shallow_wavenet_amd.runtime import current_precision
        assert current_precision() == 0
    assert rc == 0
    text = caplog.text.lower()
    assert "nan" not in text
