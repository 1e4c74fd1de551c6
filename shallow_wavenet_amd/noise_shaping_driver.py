"""Stages 3 / 6 / 9 of run.sh - counterpart of `src/bin/noise_shaping.py` (same flags): every waveform of a directory or list is
filtered with the time-invariant MLSA filter built from the corpus-mean mel-cepstrum (`--inv true`: coefficients negated - how
run.sh:529-543 applies the shaping; `--inv false`: how run.sh:725-740 restores a decoded waveform) and written as 16-bit PCM under
`--writedir`.  Arithmetic: shallow_wavenet_amd/dsp.py (csrc/swn_dsp.c); the
statistics file is read through featio (HDF5 when h5py is present, the .npz side format otherwise)."""
from __future__ import annotations

import argparse
import glob
import logging
import multiprocessing as mp
import os
import sys
from typing import Optional, Sequence

import numpy as np

from . import dsp, featio
from .decode_driver import write_wav_pcm16

FS, SHIFTMS, FFTL, MCEP_DIM_START, MCEP_ALPHA, MAG = 24000, 5.0, 1024, 5, 0.466, 0.5      # the reference's defaults


def _strtobool(v: str) -> bool:
    s = str(v).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return True
    if s in ("n", "no", "f", "false", "off", "0"):
        return False
    raise argparse.ArgumentTypeError(f"invalid truth value {v!r}")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="noise shaping / its inverse with the corpus-mean MLSA filter")
    p.add_argument("--waveforms", default=None, help="directory or list file of the input wav files")
    p.add_argument("--stats", default=None, help="statistics file (HDF5 or the .npz side format)")
    p.add_argument("--writedir", default=None, help="directory of the filtered wav files")
    p.add_argument("--fs", default=FS, type=int, help="sampling frequency")
    p.add_argument("--shiftms", default=SHIFTMS, type=float, help="frame shift in msec")
    p.add_argument("--fftl", default=FFTL, type=int, help="FFT length (kept for the reference's command lines; unused)")
    p.add_argument("--mcep_dim_start", default=MCEP_DIM_START, type=int, help="index of c(0) in the statistics vector")
    p.add_argument("--mcep_alpha", default=MCEP_ALPHA, type=float, help="all-pass constant of the mel-cepstrum")
    p.add_argument("--mag", default=MAG, type=float, help="magnification of the shaping filter")
    p.add_argument("--verbose", default=1, type=int, help="log level")
    p.add_argument("--n_jobs", default=1, type=int, help="number of worker processes")
    p.add_argument("--inv", default=False, type=_strtobool, help="inverse filtering")
    return p


def read_wav_fs(path: str):
    """(samples in [-1, 1), sampling frequency): soundfile when importable, PCM16 through scipy otherwise"""
    try:
        import soundfile as sf
        x, fs = sf.read(path)
        return np.asarray(x, dtype=np.float64), int(fs)
    except ImportError:
        from scipy.io import wavfile
        fs, x = wavfile.read(path)
        return (x.astype(np.float64) / 32768.0 if x.dtype == np.int16 else x.astype(np.float64)), int(fs)


def list_waveforms(spec: str):
    if os.path.isdir(spec):
        return sorted(glob.glob(os.path.join(spec, "**", "*.wav"), recursive=True))
    with open(spec) as f:
        return [ln.strip() for ln in f if ln.strip()]


def mean_mcep_vector(stats: str) -> np.ndarray:
    """/mean_org_lf0 of the statistics file (noise_shaping.py:166) through the lookup the training scripts use"""
    mean, _ = featio.read_stats(stats, "/feat_org_lf0")
    return np.asarray(mean, dtype=np.float64)


def shape_files(files, args, mean) -> int:
    for name in files:
        x, fs = read_wav_fs(name)
        if fs != args.fs:
            logging.error("%s: sampling frequency %d does not match --fs %d", name, fs, args.fs)
            return 1
        y = dsp.noise_shaping(x, mean, fs, args.mcep_alpha, mag=args.mag, mcep_dim_start=args.mcep_dim_start, inv=args.inv,
                              shiftms=args.shiftms)
        write_wav_pcm16(os.path.join(args.writedir, os.path.basename(name)), np.clip(y, -1.0, 1.0), fs)
        logging.info("%s -> %s", name, args.writedir)
    return 0


def _job(files, args, mean, q):
    q.put(shape_files(files, args, mean))


def main(argv: Optional[Sequence[str]] = None) -> int:
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO if args.verbose > 0 else logging.WARN, format="%(asctime)s %(message)s")
    if not args.waveforms or not args.stats or not args.writedir:
        logging.error("--waveforms, --stats and --writedir are required")
        return 2
    files = list_waveforms(args.waveforms)
    os.makedirs(args.writedir, exist_ok=True)
    mean = mean_mcep_vector(args.stats)
    if args.n_jobs <= 1 or len(files) <= 1:
        return shape_files(files, args, mean)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    parts = [list(p) for p in np.array_split(np.asarray(files, dtype=object), args.n_jobs) if len(p)]
    procs = [ctx.Process(target=_job, args=(p, args, mean, q)) for p in parts]
    for pr in procs:
        pr.start()
    rcs = [q.get() for _ in procs]
    for pr in procs:
        pr.join()
    return max(rcs) if rcs else 0


if __name__ == "__main__":
    sys.exit(main())
