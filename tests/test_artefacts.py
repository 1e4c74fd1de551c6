"""stage artefacts (model.conf, checkpoint-N.pkl): what stage 4/7 writes must be what stage 5/8 reads
(train_cswnv_laplace-stftcmplx_shift1.py:162-182,293 ; decode_cswnv_laplace-shift1.py:208-224 ; run.sh:658)."""
import argparse
import logging

import numpy as np
import pytest
import torch

from shallow_wavenet_amd import artefacts as A
from shallow_wavenet_amd import decode_driver as DD
from shallow_wavenet_amd import train_driver as T
from shallow_wavenet_amd.nets import cswnv_shift1 as mc

TINY = ["--n_aux", "10", "--hid_chn", "32", "--skip_chn", "48", "--dilation_depth", "3", "--dilation_repeat", "2",
        "--kernel_size", "3", "--upsampling_factor", "20", "--wav_conv_flag", "true"]


def test_model_conf_is_the_namespace_and_builds_a_model(tmp_path):
    args = T.build_parser().parse_args(["--expdir", str(tmp_path), "--seg", "2", "--lpc", "4"] + TINY)
    A.save_config(args, str(tmp_path / "model.conf"))
    conf = DD.load_config(str(tmp_path / "model.conf"))
    assert isinstance(conf, argparse.Namespace) and conf.n_aux == 10 and conf.string_path == "/feat_org_lf0"
    m = DD.build_model("laplace", conf)                        # attribute access, like decode_cswnv...py:208-221
    assert (m.seg, m.lpc, m.receptive_field) == (2, 4, 54)
    # round-1 files held vars(args): still accepted
    torch.save(vars(args), str(tmp_path / "old.conf"))
    assert DD.load_config(str(tmp_path / "old.conf")).hid_chn == 32


def test_checkpoint_loads_without_executing_anything_and_restores_the_numpy_stream(tmp_path):
    m = mc.CSWNV(n_aux=10, hid_chn=32, skip_chn=48, dilation_depth=3, dilation_repeat=2, kernel_size=3,
                 upsampling_factor=20, seg=1, lpc=0, wav_conv_flag=True)
    opt = torch.optim.Adam(T.optimizer_parameters(m), lr=1e-4)
    np.random.seed(5)
    np.random.rand(17)
    st = np.random.get_state()
    want = np.random.rand(4)
    T.save_checkpoint(str(tmp_path), m, opt, st, torch.get_rng_state(), 3)
    path = str(tmp_path / "checkpoint-3.pkl")
    raw = torch.load(path, weights_only=True)                 # our own files need no allow-list at all
    assert set(raw) == {"model", "optimizer", "numpy_random_state", "torch_random_state", "iterations"}
    ck = A.load_checkpoint(path)
    assert list(ck["model"]) == list(m.state_dict()) and ck["iterations"] == 3
    np.random.set_state(ck["numpy_random_state"])
    assert np.array_equal(np.random.rand(4), want)
    opt.load_state_dict(ck["optimizer"])


def test_reference_written_files_load_too(tmp_path):
    """the reference stores np.random.get_state() raw and the Namespace pickled: both allow-listed, not executed."""
    sd = {"scale_in.weight": torch.zeros(2, 2, 1)}
    torch.save({"model": sd, "optimizer": {}, "numpy_random_state": np.random.get_state(),
                "torch_random_state": torch.get_rng_state(), "iterations": 9}, str(tmp_path / "checkpoint-9.pkl"))
    ck = A.load_checkpoint(str(tmp_path / "checkpoint-9.pkl"))
    assert ck["iterations"] == 9 and ck["numpy_random_state"][1].dtype == np.uint32
    np.random.set_state(ck["numpy_random_state"])
    torch.save(argparse.Namespace(n_aux=54, string_path="/feat_org_lf0"), str(tmp_path / "model.conf"))
    assert A.load_config(str(tmp_path / "model.conf")).n_aux == 54
    with pytest.raises(RuntimeError):
        torch.save({"weights": sd}, str(tmp_path / "bad.pkl"))
        A.load_checkpoint(str(tmp_path / "bad.pkl"))


def test_checkpoint_pickled_under_numpy_1_loads(tmp_path):
    """a reference-era file names numpy's array reconstruction helper `numpy.core.multiarray._reconstruct` (numpy 1.x
    module path): rewrite the pickle stream of a checkpoint to that spelling and load it."""
    import io
    import zipfile
    buf = io.BytesIO()
    torch.save({"model": {"w": torch.ones(3)}, "optimizer": {}, "numpy_random_state": np.random.get_state(),
                "torch_random_state": torch.get_rng_state(), "iterations": 4}, buf)
    zf = zipfile.ZipFile(io.BytesIO(buf.getvalue()))
    pk = [n for n in zf.namelist() if n.endswith("data.pkl")][0]
    data = zf.read(pk)
    assert b"numpy._core.multiarray" in data
    path = tmp_path / "checkpoint-4.pkl"
    with zipfile.ZipFile(str(path), "w") as zo:
        for n in zf.namelist():
            zo.writestr(n, data.replace(b"numpy._core.multiarray", b"numpy.core.multiarray") if n == pk else zf.read(n))
    ck = A.load_checkpoint(str(path))
    assert ck["iterations"] == 4 and ck["numpy_random_state"][1].shape == (624,)


def test_empty_shard_plans_no_batches():
    """fewer utterances than ranks: the spare rank decodes nothing instead of crashing in array_split."""
    from shallow_wavenet_amd import dist as D
    shards = D.shard_utterances(["a.npy", "b.npy"], 3)
    assert [len(s) for s in shards] == [1, 1, 0]
    assert DD.plan_batches([], [], 4) == []
    assert list(DD.decode_batches(shards[2], 4, "/feat_org_lf0", 110)) == []


@pytest.mark.gpu
def test_stage4_outputs_feed_stage5_and_pretrained(gpu_ok, tmp_path):
    """train driver (one synthetic epoch) -> its expdir -> decode driver on checkpoint-1.pkl + model.conf writes
    WAVs; the same checkpoint then seeds a second training run through --pretrained and --resume."""
    import wave
    exp = tmp_path / "exp"
    common = ["--synthetic", "2", "--seg", "2", "--lpc", "4", "--batch_size", "1200", "--n_fft_facts", "5",
              "--do_prob", "0.5", "--verbose", "1"] + TINY
    assert T.main(["--expdir", str(exp), "--epoch_count", "1"] + common) == 0
    assert (exp / "checkpoint-1.pkl").exists() and (exp / "model.conf").exists()
    feats = tmp_path / "feats"
    feats.mkdir()
    rng = np.random.Generator(np.random.PCG64(3))
    for i, tf in enumerate((6, 9)):
        np.save(str(feats / f"utt{i}.npy"), rng.standard_normal((tf, 10)).astype(np.float32))
    out = tmp_path / "wav"
    rc = DD.main("laplace", ["--feats", str(feats), "--checkpoint", str(exp / "checkpoint-1.pkl"),
                             "--config", str(exp / "model.conf"), "--outdir", str(out), "--batch_size", "2"])
    assert rc == 0
    for i, tf in enumerate((6, 9)):
        with wave.open(str(out / f"utt{i}.wav")) as w:
            assert w.getnframes() == tf * 20
            pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        assert np.isfinite(pcm.astype(np.float64)).all() and np.abs(pcm).max() > 0
    assert T.main(["--expdir", str(tmp_path / "ft"), "--epoch_count", "1", "--max_iters", "2",
                   "--pretrained", str(exp / "checkpoint-1.pkl")] + common) == 0
    assert T.main(["--expdir", str(tmp_path / "rs"), "--epoch_count", "2", "--max_iters", "2",
                   "--resume", str(exp / "checkpoint-1.pkl")] + common) == 0
    logging.getLogger().handlers = [h for h in logging.getLogger().handlers if not isinstance(h, logging.FileHandler)]
