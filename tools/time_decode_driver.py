"""dev tool: the stage-8 decode driver end to end (checkpoint + model.conf from the stage-7 driver, .npy feature files,
WAV files out) at the run.sh geometry (or `bl6`): wall time against the generated audio.
  python tools/time_decode_driver.py [ref6|bl6] [n_utts] [frames]"""
import sys, os, tempfile, time, logging
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np
from shallow_wavenet_amd import train_driver as T, decode_driver as DD
shape = sys.argv[1] if len(sys.argv) > 1 else "ref6"
n_utts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 200
geom = [] if shape == "ref6" else ["--hid_chn", "64", "--skip_chn", "128", "--dilation_depth", "6", "--dilation_repeat", "1", "--kernel_size", "2"]
logging.getLogger().setLevel(logging.WARNING)
with tempfile.TemporaryDirectory() as d:
    exp = os.path.join(d, "exp")
    T.main(["--expdir", exp, "--synthetic", "2", "--epoch_count", "1", "--max_iters", "1", "--seg", "5", "--lpc", "4",
            "--wav_conv_flag", "true", "--GPU_device", "0"] + geom)
    ck = [f for f in os.listdir(exp) if f.startswith("checkpoint")][0]
    feats = os.path.join(d, "feats"); os.makedirs(feats)
    rng = np.random.Generator(np.random.PCG64(3))
    for i in range(n_utts):
        np.save(os.path.join(feats, f"utt{i:03d}.npy"), rng.standard_normal((frames - (i % 7), 54)).astype(np.float32))
    for rep in range(2):
        out = os.path.join(d, f"wav{rep}")
        t0 = time.time()
        rc = DD.main("laplace", ["--feats", feats, "--checkpoint", os.path.join(exp, ck), "--config", os.path.join(exp, "model.conf"),
                                 "--outdir", out, "--batch_size", str(n_utts)] + sys.argv[4:])
        wall = time.time() - t0
        n = sum((frames - (i % 7)) * 110 for i in range(n_utts))
        print(f"{shape} run {rep}: rc={rc} {n_utts} utterances, {n} samples ({n/22050:.1f} s of audio) in {wall:.2f} s -> {n/wall/1e3:.1f} k samples/s, {n/22050/wall:.2f}x real time")
