"""dev tool: seconds per training chunk of the stage-4 driver (train_softmax_driver.main) on synthetic utterances,
run.sh-like flags.   python tools/time_softmax_driver.py [bf16|fp32] [iters] [extra driver flags]"""
import sys, os, re, tempfile, logging, io, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
from shallow_wavenet_amd import train_softmax_driver as T
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
buf = io.StringIO()
h = logging.StreamHandler(buf); logging.getLogger().addHandler(h); logging.getLogger().setLevel(logging.INFO)
with tempfile.TemporaryDirectory() as d:
    t0 = time.time()
    T.main(["--expdir", d, "--synthetic", "6", "--max_iters", str(iters), "--n_aux", "54", "--hid_chn", "256", "--skip_chn", "256",
            "--dilation_depth", "3", "--dilation_repeat", "2", "--kernel_size", "7", "--batch_size", "8800", "--do_prob", "0.5",
            "--precision", prec, "--GPU_device", "0", "--verbose", "1"] + sys.argv[3:])
    wall = time.time() - t0
secs = [float(m) for m in re.findall(r"\((\d+\.\d+) sec\)", buf.getvalue())]
tail = secs[5:] if len(secs) > 10 else secs
print(f"{prec}: {len(secs)} chunks, median {sorted(tail)[len(tail)//2]*1e3:.1f} ms per chunk (first {secs[0]*1e3:.0f} ms), wall {wall:.1f} s") if secs else print(buf.getvalue()[-1500:])
