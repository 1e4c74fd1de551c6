"""dev tool: cProfile of the stage-7 training driver over synthetic chunks (host view)."""
import sys, os, tempfile, cProfile, pstats, logging
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
from shallow_wavenet_amd import train_driver as T
logging.getLogger().setLevel(logging.WARNING)
with tempfile.TemporaryDirectory() as d:
    args = ["--expdir", d, "--synthetic", "6", "--seg", "5", "--lpc", "4", "--do_prob", "0.5", "--wav_conv_flag", "true",
            "--precision", "bf16", "--GPU_device", "0", "--verbose", "1"]
    T.main(args + ["--max_iters", "8"])          # warm-up (library load, allocator)
    pr = cProfile.Profile(); pr.enable()
    T.main(args + ["--max_iters", "30"])
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(38)
