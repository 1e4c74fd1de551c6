"""decode drivers: CPU known-answer tests of batching / padding / WAV writing /
CLI, and a GPU end-to-end run that must equal the module API and the reference fixture."""
import json
import os
import subprocess
import sys
import wave

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from shallow_wavenet_amd import config as C
from shallow_wavenet_amd import decode_driver as DD
from shallow_wavenet_amd.synth import synth_state_dict


def test_plan_batches_known_answer():
    """decode_generator (decode_cswnv...py:77-84): argsort by frames, ceil(N/bs) near-equal cuts."""
    files = [f"u{i}.npy" for i in range(7)]
    frames = [50, 10, 40, 20, 70, 30, 60]
    got = DD.plan_batches(files, frames, 3)
    assert got == [["u1.npy", "u3.npy", "u5.npy"], ["u2.npy", "u0.npy"], ["u6.npy", "u4.npy"]]
    assert DD.plan_batches(files, frames, 7) == [["u1.npy", "u3.npy", "u5.npy", "u2.npy", "u0.npy", "u6.npy", "u4.npy"]]
    assert [len(b) for b in DD.plan_batches(files, frames, 1)] == [1] * 7


def test_pad_list_zero_pads_raw_features():
    a, b = np.ones((3, 2)), 2 * np.ones((5, 2))
    p = DD.pad_list([a, b])
    assert p.shape == (2, 5, 2) and p.dtype == np.float64
    assert np.array_equal(p[0, :3], a) and np.all(p[0, 3:] == 0) and np.array_equal(p[1], b)


def test_wav_writer_matches_libsndfile_pcm16_rule(tmp_path):
    x = np.array([0.0, 1.0, -1.0, 0.5, -0.25, 1e-5, 0.99999])
    path = str(tmp_path / "a.wav")
    DD.write_wav_pcm16(path, x, 22050)
    with wave.open(path) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 22050, 7)
        pcm = np.frombuffer(w.readframes(7), dtype="<i2")
    assert pcm.tolist() == [0, 32767, -32767, 16384, -8192, 0, 32767]      # lrint(x * 0x7FFF)


def test_feature_listing_and_reading(tmp_path):
    d = tmp_path / "feats" / "spk"
    d.mkdir(parents=True)
    for i, t in enumerate((4, 9)):
        np.save(str(d / f"utt{i}.npy"), np.full((t, 3), i, dtype=np.float32))
    files = DD.list_features(str(tmp_path / "feats"))
    assert [os.path.basename(f) for f in files] == ["utt0.npy", "utt1.npy"]
    assert [DD.feature_frames(f, "/x") for f in files] == [4, 9]
    lst = tmp_path / "list.scp"
    lst.write_text("\n".join(files) + "\n")
    assert DD.list_features(str(lst)) == files
    batches = list(DD.decode_batches(files, 2, "/x", 10))
    assert batches[0][0] == ["utt0", "utt1"] and batches[0][1].shape == (2, 9, 3) and batches[0][2] == [40, 90]
    with pytest.raises(FileNotFoundError):
        DD.list_features(str(tmp_path / "nope"))


def test_cli_flags_are_the_reference_flags():
    p = DD.make_parser()
    a = p.parse_args("--feats f --checkpoint c --config m --outdir o --fs 16000 --batch_size 7 --n_gpus 3 "
                     "--intervals 100 --seed 2 --GPU_device 1 --GPU_device_str 1,2,0 --verbose 0".split())
    assert (a.fs, a.batch_size, a.n_gpus, a.intervals, a.seed, a.GPU_device, a.GPU_device_str) == \
        (16000, 7, 3, 100, 2, 1, "1,2,0")
    assert a.spk_trg is None and a.min_idx is None


def _write_run(tmp_path, cfg, d, kind):
    feats = tmp_path / "feats"
    feats.mkdir()
    names = []
    for b, f in enumerate(d["frames"]):
        np.save(str(feats / f"utt{b}.npy"), d["aux"][b, :, : int(f)].T.astype(np.float32))
        names.append(f"utt{b}")
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"])).items()}
    torch.save({"model": sd}, str(tmp_path / "checkpoint-1.pkl"))
    conf = dict(cfg.to_dict(), string_path="/feat_org_lf0", audio_in=cfg.audio_in_flag)
    (tmp_path / "model.json").write_text(json.dumps(conf))
    return feats, names


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind", [("g0_tiny_lap_s2l4_trained", "laplace"), ("g0_tiny_softmax", "softmax")])
def test_driver_end_to_end_reproduces_the_reference_batch(gpu_ok, tmp_path, name, kind):
    """features on disk -> driver CLI -> WAV files; one batch holding both utterances, seed = the
    fixture's noise seed: the PCM must equal the reference's samples quantised by the same rule."""
    cfg, d = load_golden(name)
    feats, names = _write_run(tmp_path, cfg, d, kind)
    out = tmp_path / "wav"
    script = os.path.join(ROOT, "shallow_wavenet_amd", "bin",
                          "decode_cswnv_laplace-shift1.py" if kind == "laplace" else "decode_dswnv_softmax.py")
    r = subprocess.run([sys.executable, script, "--feats", str(feats), "--checkpoint", str(tmp_path / "checkpoint-1.pkl"),
                        "--config", str(tmp_path / "model.json"), "--outdir", str(out), "--fs", "22050",
                        "--batch_size", "2", "--seed", str(int(d["noise_seed"])), "--verbose", "1",
                        "--noise_source", "host"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "average throughput / sample" in r.stderr + open(out / "decode.log").read()
    # the driver sorts the batch by ascending frame count; decode the same batch with the pinned oracle
    from oracle import cpu_ref
    order = np.argsort([int(f) for f in d["frames"]])
    aux = np.zeros_like(d["aux"])
    for pos, b in enumerate(order):
        aux[pos, :, : int(d["frames"][b])] = d["aux"][b, :, : int(d["frames"][b])]
    n_samples = [int(d["n_samples"][b]) for b in order]
    P = cpu_ref.as_params(synth_state_dict(cfg, seed=int(d["wseed"]), flavor=str(d["flavor"])))
    # replay the driver's RNG use: seed, construct the module (default init draws from the same
    # generator, as in the reference script), then the noise
    from types import SimpleNamespace
    np.random.seed(int(d["noise_seed"]))
    torch.manual_seed(int(d["noise_seed"]))
    DD.build_model(kind, SimpleNamespace(**dict(cfg.to_dict(), audio_in=cfg.audio_in_flag)))
    g = None
    if kind == "laplace":
        noise = cpu_ref.laplace_noise(cfg, max(n_samples) // cfg.seg, len(order), generator=g)
        ref = cpu_ref.laplace_generate(cfg, P, torch.from_numpy(aux), n_samples, noise)
    else:
        from shallow_wavenet_amd.nets.dswnv import decode_mu_law
        noise = cpu_ref.softmax_noise(cfg, max(n_samples), len(order), generator=g)
        ref = [decode_mu_law(r, cfg.n_quantize) for r in cpu_ref.softmax_generate(cfg, P, torch.from_numpy(aux), n_samples, noise)]
    for pos, b in enumerate(order):
        with wave.open(str(out / f"{names[b]}.wav")) as w:
            assert w.getnframes() == n_samples[pos] and w.getframerate() == 22050
            pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.int64)
        want = np.rint(np.clip(ref[pos], -1, 1).astype(np.float64) * 32767.0).astype(np.int64)
        assert np.abs(pcm - want).max() <= (1 if kind == "laplace" else 0), (name, pos)


def test_listing_dedupes_formats_and_skips_non_utterance_containers(tmp_path):
    """an utterance present as x.npz and x.npy is decoded once; calc_stats' stats.npz in the tree is not an utterance."""
    from shallow_wavenet_amd import featio
    d = tmp_path / "feats"
    d.mkdir()
    a = np.arange(12, dtype=np.float32).reshape(4, 3)
    np.save(str(d / "utt0.npy"), a)
    featio.write_dataset(str(d / "utt0.npz"), "/feat_org_lf0", a)
    featio.write_dataset(str(d / "utt1.npz"), "/feat_org_lf0", a)
    featio.write_stats(str(d / "stats.npz"), "/feat_org_lf0", a.mean(0), a.std(0) + 1)
    got = DD.list_features(str(d), "/feat_org_lf0")
    assert [os.path.basename(f) for f in got] == ["utt0.npz", "utt1.npz"]
    assert "stats.npz" in [os.path.basename(f) for f in DD.list_features(str(d))]       # without a dataset name: no probing


def _tiny_run(tmp_path, kind, frames):
    cfg = C.tiny(kind, 2, 4) if kind == "laplace" else C.tiny("softmax", wav_conv_flag=False)
    feats = tmp_path / "feats"
    feats.mkdir()
    rng = np.random.default_rng(3)
    for i, f in enumerate(frames):
        np.save(str(feats / f"utt{i:02d}.npy"), rng.standard_normal((f, cfg.n_aux)).astype(np.float32))
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed=7, flavor="trained" if kind == "laplace" else "xavier").items()}
    torch.save({"model": sd}, str(tmp_path / "checkpoint-1.pkl"))
    (tmp_path / "model.json").write_text(json.dumps(dict(cfg.to_dict(), string_path="/feat_org_lf0", audio_in=cfg.audio_in_flag)))
    return cfg, ["--feats", str(feats), "--checkpoint", str(tmp_path / "checkpoint-1.pkl"), "--config", str(tmp_path / "model.json"),
                 "--fs", "22050", "--verbose", "0"]


def _clean_env(monkeypatch):
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        monkeypatch.delenv(k, raising=False)


def test_n_gpus_fans_out_its_own_ranks(tmp_path, monkeypatch):
    """`--n_gpus 2` as run.sh:675-684 passes it, no launcher around it: main() spawns the two ranks (the parent touches no
    GPU), they meet in the one broadcast over gloo and each plans its np.array_split shard with global utterance indices."""
    _clean_env(monkeypatch)
    frames = [5, 3, 4, 6, 2]
    cfg, argv = _tiny_run(tmp_path, "laplace", frames)
    out = tmp_path / "plan"
    rc = DD.main("laplace", argv + ["--outdir", str(out), "--n_gpus", "2", "--batch_size", "2", "--plan_only"])
    assert rc == 0
    plans = [json.load(open(out / f"decode.{r}.plan.json")) for r in range(2)]
    assert [p["index"] for p in plans] == [[0, 1, 2], [3, 4]]                       # np.array_split of the unsorted list
    assert [os.path.basename(f) for f in plans[1]["shard"]] == ["utt03.npy", "utt04.npy"]
    assert plans[0]["rng_key"] == plans[1]["rng_key"]                               # one generator key for the whole run
    assert plans[0]["packed_sum"] == plans[1]["packed_sum"] and plans[0]["packed_numel"] == plans[1]["packed_numel"] > 0
    # rank 0: frames 5,3,4 sorted -> [utt01, utt02 | utt00]; each utterance keeps its position in the full list
    assert plans[0]["batches"] == [{"ids": ["utt01", "utt02"], "n_samples": [3 * cfg.U, 4 * cfg.U], "utt_index": [1, 2]},
                                   {"ids": ["utt00"], "n_samples": [5 * cfg.U], "utt_index": [0]}]
    assert plans[1]["batches"] == [{"ids": ["utt04", "utt03"], "n_samples": [2 * cfg.U, 6 * cfg.U], "utt_index": [4, 3]}]
    # the single-process plan covers the same utterances with the same indices and the same key
    rc = DD.main("laplace", argv + ["--outdir", str(tmp_path / "plan1"), "--n_gpus", "1", "--batch_size", "2", "--plan_only"])
    assert rc == 0
    solo = json.load(open(tmp_path / "plan1" / "decode.0.plan.json"))
    assert solo["index"] == [0, 1, 2, 3, 4] and solo["rng_key"] == plans[0]["rng_key"]
    pair = {i: u for p in plans for b in p["batches"] for i, u in zip(b["ids"], b["utt_index"])}
    assert pair == {i: u for b in solo["batches"] for i, u in zip(b["ids"], b["utt_index"])}


def test_a_failing_rank_fails_the_fan_out(tmp_path, monkeypatch):
    _clean_env(monkeypatch)
    _, argv = _tiny_run(tmp_path, "laplace", [3, 2])
    argv[argv.index("--checkpoint") + 1] = str(tmp_path / "missing.pkl")              # rank 0 cannot load it
    rc = DD.main("laplace", argv + ["--outdir", str(tmp_path / "o"), "--n_gpus", "2", "--plan_only"])
    assert rc != 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["laplace", "softmax"])
def test_device_noise_does_not_depend_on_n_gpus(gpu_ok, tmp_path, monkeypatch, kind):
    """the same list decoded by one process and by `--n_gpus 2` (two spawned ranks; on a one-GPU box they share the card and
    the broadcast runs over gloo) with `--noise_source device`: identical WAV files - an utterance's noise stream is keyed by
    the run's one key and its position in the unsorted list, not by its shard or by where the length sort puts it."""
    _clean_env(monkeypatch)
    frames = [4, 2, 5, 3, 6]
    cfg, argv = _tiny_run(tmp_path, kind, frames)
    outs = []
    for n in (1, 2):
        out = tmp_path / f"wav{n}"
        rc = DD.main(kind, argv + ["--outdir", str(out), "--n_gpus", str(n), "--batch_size", "1", "--seed", "5",
                                   "--noise_source", "device"])
        assert rc == 0
        outs.append(out)
    for i, f in enumerate(frames):
        a, b = (open(o / f"utt{i:02d}.wav", "rb").read() for o in outs)
        assert len(a) == 44 + 2 * f * cfg.U and a == b, (kind, i)
    assert (outs[1] / "decode.0.log").exists() and (outs[1] / "decode.1.log").exists()
    # and the streams differ between utterances (the index really enters the generator)
    assert open(outs[0] / "utt00.wav", "rb").read()[44:200] != open(outs[0] / "utt03.wav", "rb").read()[44:200]
