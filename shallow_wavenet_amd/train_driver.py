"""Stage-4 training driver for the Laplace model (row f2 of SURVEY.md section 8): the host logic of the
reference's `src/bin/train_cswnv_laplace-stftcmplx_shift1.py` around the HIP forward / backward.

Restated here (reference file:line in each docstring): the chunk generator, the slicing of one chunk into
network input / LP context / target, the batch loss (Laplace NLL + reparameterised-sample complex-STFT L1, with
the log-spectral distance and the sample error as reported figures), the optimizer parameter list, the checkpoint
dictionary and the log lines the recipe's awk scripts parse.  The network itself is `nets/cswnv_shift1.CSWNV`:
its forward and backward run as HIP kernels; only the loss arithmetic on the network's outputs (a few
element-wise ops and torch.stft on the device) is ordinary torch, exactly as in the reference.

File formats: wav through `soundfile` when present, else PCM16 through scipy; features / statistics through
`h5py` when present, else `.npy` / `.npz` files of the same stem (h5py is absent in this image).
`--synthetic N` trains on N generated utterances without touching the disk (smoke runs and timing).
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
import time
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import artefacts, featio

FFT_FACTS = {5: [128, 256, 512, 1024, 2048],
             9: [128, 192, 256, 384, 512, 768, 1024, 1536, 2048],
             17: [128, 160, 192, 224, 256, 320, 384, 448, 512, 640, 768, 896, 1024, 1280, 1536, 1792, 2048]}


def fft_sizes(n_fft_facts: int) -> List[int]:
    """STFT sizes of the spectral loss (train_cswnv...py:452-476)."""
    if n_fft_facts in FFT_FACTS:
        return list(FFT_FACTS[n_fft_facts])
    out, init_fft = [], 64
    for i in range(n_fft_facts):
        if i % 2 == 0:
            init_fft *= 2
            out.append(init_fft)
        else:
            out.append(init_fft + init_fft // 2)
    return out


# ----------------------------------------------------------------------------------------- data
def validate_length(x: np.ndarray, h: np.ndarray, upsampling_factor: int) -> Tuple[np.ndarray, np.ndarray]:
    """trim waveform and features to len(x) == len(h) * U (train_cswnv...py:38-66)."""
    if upsampling_factor == 0:
        n = min(x.shape[0], h.shape[0])
        return x[:n], h[:n]
    mod = x.shape[0] % upsampling_factor
    if mod > 0:
        x = x[:-mod]
    if x.shape[0] > h.shape[0] * upsampling_factor:
        x = x[:-(x.shape[0] - h.shape[0] * upsampling_factor)]
    elif x.shape[0] < h.shape[0] * upsampling_factor:
        h = h[:-((h.shape[0] * upsampling_factor - x.shape[0]) // upsampling_factor)]
    assert len(x) == len(h) * upsampling_factor
    return x, h


def effective_batch_size(receptive_field: int, batch_size: int, seg: int, upsampling_factor: int) -> int:
    """the reference shrinks batch_size so that rf + batch_size + seg ... is cut on frame boundaries
    (train_cswnv...py:95-102; note: `chunk` itself keeps the ORIGINAL batch_size)."""
    chunk = receptive_field + batch_size + seg
    if batch_size != 0 and upsampling_factor != 0:
        batch_size -= chunk % upsampling_factor
    return batch_size


def chunk_plan(n_frames: int, receptive_field: int, batch_size: int, seg: int, upsampling_factor: int
               ) -> List[Tuple[int, int, int, int]]:
    """(h_bs, x_bs, h_ss, x_ss) of every chunk the reference yields for an utterance of n_frames frames
    (train_cswnv...py:96,126-145): full chunks of `chunk` samples advancing by batch_size//U frames, then one
    open-ended tail chunk (h_bs = x_bs = -1) if at least seg samples are left beyond rf + seg."""
    U = upsampling_factor
    chunk = receptive_field + batch_size + seg
    bs = effective_batch_size(receptive_field, batch_size, seg, U)
    h_bs = chunk // U
    x_bs = h_bs * U
    delta = bs // U
    out, len_frm, h_ss, x_ss = [], n_frames, 0, 0
    while True:
        if len_frm * U - chunk >= seg:
            out.append((h_bs, x_bs, h_ss, x_ss))
            h_ss += delta
            x_ss = h_ss * U
            len_frm -= delta
        elif len_frm * U - (receptive_field + seg) >= seg:
            out.append((-1, -1, h_ss, x_ss))
            break
        else:
            break
    return out


def make_adam(params, lr: float):
    """torch.optim.Adam as the reference constructs it (train_cswnv...py:456: default betas / eps, no weight decay), in torch's single-
    launch `fused` form when every parameter lives on a GPU: the same update rule in one kernel instead of four multi-tensor passes
    over the 51 parameter tensors (BL6 step 1.49 -> 1.36 ms; the optimizer's state_dict is interchangeable with the default form)."""
    params = list(params)
    flat = [q for p in params for q in (p["params"] if isinstance(p, dict) else [p])]
    fused = len(flat) > 0 and all(q.is_cuda and q.is_floating_point() for q in flat)
    return torch.optim.Adam(params, lr=lr, **({"fused": True} if fused else {}))


def read_wav(path: str) -> np.ndarray:
    try:
        import soundfile as sf
        x, _ = sf.read(path)
        return np.asarray(x, dtype=np.float64)
    except ImportError:
        from scipy.io import wavfile
        _, x = wavfile.read(path)
        if x.dtype == np.int16:
            return x.astype(np.float64) / 32768.0
        return x.astype(np.float64)


def read_feat(path: str, string_path: str) -> np.ndarray:
    """(Tf, n_aux) features: HDF5 dataset `string_path` (utils.py read_hdf5) or its .npz / .npy side file (featio.py)."""
    return np.asarray(featio.read_dataset(featio.resolve(path), string_path))


def read_stats(path: str, string_path: str) -> Tuple[np.ndarray, np.ndarray]:
    """mean / scale of the features -> scale_in initialisation (train_cswnv...py:316-345)."""
    return featio.read_stats(path, string_path)


def train_generator(wav_list: Sequence, feat_list: Sequence, receptive_field: int, string_path: str = "/feat_org_lf0",
                    batch_size: int = 8800, seg: int = 1, training: bool = True, upsampling_factor: int = 110,
                    device: Optional[torch.device] = None, loader=None) -> Iterator:
    """the reference's chunk generator (train_cswnv...py:69-159): yields
    (x, h, c_idx, utt_idx, wavfile, h_bs, x_bs, h_ss, x_ss) per chunk and a c_idx = -1 record at the end of
    every epoch; utterances are reshuffled per epoch with np.random.permutation when training.
    `loader(wav, feat) -> (x, h)` replaces the file readers (synthetic data)."""
    n_files = len(wav_list)
    idx = np.random.permutation(n_files) if training else np.arange(n_files)
    while True:
        for c_idx, i in enumerate(idx):
            wavfile, featfile = wav_list[i], feat_list[i]
            if loader is not None:
                x, h = loader(wavfile, featfile)
            else:
                x, h = read_wav(wavfile), read_feat(featfile, string_path)
            x, h = validate_length(np.asarray(x), np.asarray(h), upsampling_factor)
            xt = torch.as_tensor(x, dtype=torch.float32, device=device)
            ht = torch.as_tensor(h, dtype=torch.float32, device=device)
            for h_bs, x_bs, h_ss, x_ss in chunk_plan(len(h), receptive_field, batch_size, seg, upsampling_factor):
                yield xt, ht, c_idx, int(i), wavfile, h_bs, x_bs, h_ss, x_ss
        yield [], [], -1, -1, [], [], [], [], []
        if training:
            idx = np.random.permutation(n_files)


# ----------------------------------------------------------------------------------------- one chunk
def slice_chunk(model, x_float: torch.Tensor, h: torch.Tensor, h_bs: int, x_bs: int, h_ss: int, x_ss: int):
    """one generator record -> (batch_h (1,n_aux,Tf'), batch_x (1,1,T'), target (T''), lp context or None, feat_len)
    (train_cswnv...py:702-742)."""
    seg, lpc, off, rf = model.seg, model.lpc, model.lpc_offset, model.receptive_field
    batch_h = h[h_ss:]
    x_ = x_float[x_ss:]
    x_lpc = None
    if lpc > 0:
        x_lpc = x_float[x_ss + off:] if x_ss + off >= 0 else x_float[x_ss:]
    x_prob = None
    if h_bs != -1:
        batch_h = batch_h[:h_bs]
        if lpc > 0:
            if x_ss + off >= 0:
                x_prob = x_lpc[:x_bs - off].unsqueeze(0)
            else:
                x_prob = F.pad(x_lpc[:x_bs], (-(x_ss + off), 0), "constant", 0).unsqueeze(0)
        batch_x, target = x_[:x_bs - seg], x_[seg:x_bs]
    else:
        if lpc > 0:
            if x_ss + off > 0:
                x_prob = x_lpc.unsqueeze(0)
            else:
                x_prob = F.pad(x_lpc, (-(x_ss + off), 0), "constant", 0).unsqueeze(0)
        batch_x, target = x_[:-seg], x_[seg:]
    batch_h = batch_h.transpose(0, 1).unsqueeze(0)
    batch_x = batch_x.unsqueeze(0).unsqueeze(1)
    if h_ss > 0:
        feat_len = (target[rf:-(seg - 1)] if seg > 1 else target[rf:]).shape[0]
    else:
        feat_len = (target[:-(seg - 1)] if seg > 1 else target).shape[0]
    return batch_h, batch_x, target, x_prob, feat_len


def _stft(x: torch.Tensor, n_fft: int, window: torch.Tensor) -> torch.Tensor:
    """the (freq, frames, 2) real view `torch.stft` returned when the reference was written."""
    return torch.view_as_real(torch.stft(x, n_fft, window=window, return_complex=True))


def batch_loss(model, criterion_laplace, criterion_lsd, batch_h, batch_x, target, x_prob, feat_len, h_ss,
               fft_facts: Sequence[int], hann_win: Sequence[torch.Tensor], do: bool = True,
               eps_generator: Optional[torch.Generator] = None, eps_on_device: bool = False):
    """forward + loss of one chunk (train_cswnv...py:744-868): returns
    (batch_loss, batch_loss_laplace, batch_loss_lsd or None, batch_loss_err).

    mean of the LP part: a = flip(a); for j < seg: mu_j += sum_k a_k * x[t+j-lpc+k] by `unfold` on the LP context;
    loss = mean_j NLL_j + mean_j mean_fft L1(STFT(sample_j), STFT(target_j)) with
    sample = mu - b_noclip * sign(eps) * log1p(-2|eps|), eps ~ U(-0.4999, 0.5): drawn on the HOST generator by default (what
    the g6_trainstep fixtures replay), with eps_on_device=True on the model's device like the reference (the driver does
    that whenever its dropout masks are device-drawn: a host draw is a pageable host->device copy per segment)."""
    seg, lpc, rf = model.seg, model.lpc, model.receptive_field
    if lpc > 0:
        mus, bs_noclip, bs, log_bs, ass = model(batch_h, batch_x, do=do, clip=True)
        ass = ass.flip(-1)
        init_mus = mus
        for j in range(seg):
            tmp = x_prob[:, j:-(seg - j)].unfold(1, lpc, 1)
            lp = torch.sum(ass * tmp, -1, keepdim=True)
            mus = lp + init_mus[:, :, j:j + 1] if j == 0 else torch.cat((mus, lp + init_mus[:, :, j:j + 1]), 2)
        if seg == 1:
            mus, bs_noclip, bs, log_bs = (t.reshape(t.shape[0], -1) for t in (mus, bs_noclip, bs, log_bs))
    else:
        mus, bs_noclip, bs, log_bs = model(batch_h, batch_x, do=do, clip=True)
    if h_ss > 0:
        mus, bs_noclip, bs, log_bs, target = mus[0, rf:], bs_noclip[0, rf:], bs[0, rf:], log_bs[0, rf:], target[rf:]
    else:
        mus, bs_noclip, bs, log_bs = mus[0], bs_noclip[0], bs[0], log_bs[0]

    def draw(shape):
        if eps_on_device:
            return torch.empty(shape, device=mus.device).uniform_(-0.4999, 0.5)
        return torch.empty(shape).uniform_(-0.4999, 0.5, generator=eps_generator).to(mus.device)

    # The spectral terms (train_cswnv...py:786-868).  The reference runs, per segment j and per FFT size, two `torch.stft`s
    # and two LSDloss calls and keeps a term only `if not torch.isinf(v) and not torch.isnan(v)`: 340 small STFTs, ~2 400
    # tiny launches and 170 device->host round trips per chunk at run.sh's seg = 5 / 17 sizes - on this GPU that, not the
    # network (2-3 ms), is the chunk time (34 ms).  Same numbers with the segments batched: per FFT size ONE stft over
    # the 2 x seg stacked signals (rows are independent), the two LSDloss formulas (cswnv_shift1.py:456-476) evaluated per
    # row, and the finite-term selection as masks on the device: a term that is not finite contributes 0 to the sum and 0
    # to the count; mean = sum / count, "no term at all" (the reference's empty list) is count == 0.
    def spectral(samples, targets):
        """samples / targets: lists of R equally long signals -> ((l1_mean[R], l1_count[R]), (lsd_mean[R], lsd_count[R]))
        or (None, None) when no FFT size applies"""
        R = len(samples)
        sig = torch.stack(list(samples) + list(targets))                      # (2R, T)
        l1, lsd = [], []
        for n_fft, win in zip(fft_facts, hann_win):
            if feat_len > n_fft // 2:
                sp = _stft(sig, n_fft, win)                                   # (2R, F, N, 2)
                so, st = sp[:R], sp[R:]
                l1.append(torch.abs(so - st).mean(dim=(1, 2, 3)))             # LSDloss(LSD=False, L2=False)
                pow_x, pow_y = torch.sum(so ** 2, -1), torch.sum(st ** 2, -1)
                lsd.append(torch.sqrt(torch.mean((10 * (torch.log10(pow_x) - torch.log10(pow_y))) ** 2, 1)).mean(1))   # LSDloss()
        if not l1:
            return None, None

        def masked_mean(rows):                                                # [K] x (R,) -> mean over the finite of the K terms
            v = torch.stack(rows, 1)                                          # (R, K)
            ok = torch.isfinite(v)
            n = ok.sum(1)
            return torch.where(ok, v, torch.zeros_like(v)).sum(1) / n.clamp(min=1), n
        return masked_mean(l1), masked_mean(lsd)

    def mean_of_present(mean_r, count_r):
        """mean over the segments whose own list was not empty (count > 0), like the reference's `if l1:` append"""
        has = count_r > 0
        k = has.sum()
        return torch.where(has, mean_r, torch.zeros_like(mean_r)).sum() / k.clamp(min=1), k

    if seg > 1:
        nll, err, samples, targets = [], [], [], []
        for i in range(seg):
            mus_i, bn_i = mus[:, i], bs_noclip[:, i]
            trg_i = target[i:-(seg - (i + 1))] if i + 1 < seg else target[i:]
            nll.append(criterion_laplace(mus_i, bs[:, i], trg_i, log_b=log_bs[:, i], log=(i == 0)))
            eps = draw(mus_i.shape)
            sample = mus_i - bn_i * eps.sign() * torch.log1p(-2 * eps.abs())
            err.append(torch.mean(torch.abs(sample - trg_i)))
            samples.append(sample); targets.append(trg_i)
        loss_laplace = torch.mean(torch.stack(nll))
        loss_err = torch.mean(torch.stack(err))
    else:
        loss_laplace = criterion_laplace(mus, bs, target, log_b=log_bs)
        eps = draw(mus.shape)
        sample = mus - bs_noclip * eps.sign() * torch.log1p(-2 * eps.abs())
        loss_err = torch.mean(torch.abs(sample - target))
        samples, targets = [sample], [target]
    l1_terms, lsd_terms = spectral(samples, targets)
    loss, loss_lsd = loss_laplace, None
    if l1_terms is not None:
        loss = loss_laplace + mean_of_present(*l1_terms)[0]                 # no finite term anywhere: + 0, like `else 0`
        lsd_mean, lsd_k = mean_of_present(*lsd_terms)
        # "no LSD figure" (every term not finite) is reported as None: the one host decision left, taken where the caller
        # is about to call .item() for its log line anyway
        loss_lsd = lsd_mean if int(lsd_k.item()) > 0 else None
    return loss, loss_laplace, loss_lsd, loss_err


def optimizer_parameters(model) -> List[torch.nn.Parameter]:
    """everything but the frozen `scale_in`, in the reference's order (train_cswnv...py:355-365)."""
    mods = [model.conv_aux, model.upsampling]
    if model.aux_conv2d_flag and model.seg > 1:
        mods.append(model.aux_conv2d)
    if model.wav_conv_flag:
        mods.append(model.wav_conv)
    mods += [model.causal, model.in_x, model.dil_h, model.out_skip, model.out_1, model.out_2]
    return [p for m in mods for p in m.parameters()]


def set_scale_in(model, mean: np.ndarray, scale: np.ndarray) -> None:
    """scale_in = diag(1/sigma), bias -mu/sigma, frozen (train_cswnv...py:344-350)."""
    dev = next(model.parameters()).device
    mean_t = torch.as_tensor(mean, dtype=torch.float32, device=dev)
    std_t = torch.as_tensor(scale, dtype=torch.float32, device=dev)
    model.scale_in.weight = torch.nn.Parameter(torch.unsqueeze(torch.diag(1.0 / std_t), 2))
    model.scale_in.bias = torch.nn.Parameter(-(mean_t / std_t))
    for p in model.parameters():
        p.requires_grad = True
    for p in model.scale_in.parameters():
        p.requires_grad = False


def save_checkpoint(checkpoint_dir: str, model, optimizer, numpy_random_state, torch_random_state, iterations: int):
    """same dictionary as the reference (train_cswnv...py:162-182), loadable with weights_only=True (artefacts.py)."""
    artefacts.save_checkpoint(checkpoint_dir, model, optimizer, numpy_random_state, torch_random_state, iterations)
    logging.info("%d-iter checkpoint created." % iterations)


# ----------------------------------------------------------------------------------------- synthetic data
def synthetic_corpus(n_utts: int, n_aux: int, upsampling_factor: int, min_frames: int = 100, max_frames: int = 140,
                     seed: int = 0):
    """in-memory utterances: N(0,1) features and a band-limited-ish noise waveform in [-1, 1]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    data = {}
    for i in range(n_utts):
        tf = int(rng.integers(min_frames, max_frames + 1))
        h = rng.standard_normal((tf, n_aux))
        x = np.tanh(0.3 * np.convolve(rng.standard_normal(tf * upsampling_factor), np.ones(8) / 8.0, mode="same"))
        data[f"synthetic/utt{i:04d}.wav"] = (x, h)
    names = sorted(data)
    return names, names, (lambda wav, feat: data[wav])


# ----------------------------------------------------------------------------------------- main
def build_parser() -> argparse.ArgumentParser:
    """flags of train_cswnv...py:185-249 (+ --synthetic / --max_iters for runs without a corpus)."""
    sb = lambda v: str(v).lower() in ("1", "true", "yes", "y", "t", "on")
    p = argparse.ArgumentParser()
    p.add_argument("--waveforms", type=str)
    p.add_argument("--waveforms_eval", type=str)
    p.add_argument("--feats", type=str)
    p.add_argument("--feats_eval", type=str)
    p.add_argument("--stats", type=str)
    p.add_argument("--expdir", required=True, type=str)
    p.add_argument("--n_aux", default=54, type=int)
    p.add_argument("--skip_chn", default=256, type=int)
    p.add_argument("--seg", default=1, type=int)
    p.add_argument("--dilation_depth", default=3, type=int)
    p.add_argument("--dilation_repeat", default=2, type=int)
    p.add_argument("--hid_chn", default=192, type=int)
    p.add_argument("--kernel_size", default=7, type=int)
    p.add_argument("--aux_kernel_size", default=3, type=int)
    p.add_argument("--aux_dilation_size", default=2, type=int)
    p.add_argument("--upsampling_factor", default=110, type=int)
    p.add_argument("--n_fft_facts", default=17, type=int)
    p.add_argument("--string_path", default="/feat_org_lf0", type=str)
    p.add_argument("--lr", default=1e-4, type=float)
    p.add_argument("--batch_size", default=8800, type=int)
    p.add_argument("--epoch_count", default=4000, type=int)
    p.add_argument("--do_prob", default=0, type=float)
    p.add_argument("--lpc", default=0, type=int)
    p.add_argument("--aux_conv2d_flag", default=False, type=sb)
    p.add_argument("--wav_conv_flag", default=False, type=sb)
    p.add_argument("--seed", default=1, type=int)
    p.add_argument("--resume", default=None, type=str)
    p.add_argument("--pretrained", default=None, type=str)
    p.add_argument("--GPU_device", default=None, type=int)
    p.add_argument("--verbose", default=1, type=int)
    p.add_argument("--synthetic", default=0, type=int, help="train on N generated utterances (no files needed)")
    p.add_argument("--max_iters", default=0, type=int, help="stop after this many chunks (0 = run all epochs)")
    p.add_argument("--dropout_source", default="device", choices=["device", "host"],
                   help="where the nn.Dropout masks (and the loss's reparameterisation noise) are drawn: the model's device (like "
                        "the reference on a GPU) or the torch CPU generator in the reference's order (reproduces a CPU run of "
                        "the reference; ~0.1 s per chunk, ~1 s at cfg4's batch)")
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                   help="arithmetic of the training step's contractions (not a reference flag): fp32 = parity mode, "
                        "bf16 = bf16 operands with fp32 accumulation (SWN_PRECISION_BF16)")
    return p


def _file_lists(wavs: str, feats: str):
    if os.path.isdir(wavs):
        names = sorted(f for f in os.listdir(wavs) if f.endswith(".wav"))
        return [os.path.join(wavs, n) for n in names], [os.path.join(feats, n.replace(".wav", ".h5")) for n in names]
    with open(wavs) as f:
        w = [ln.strip() for ln in f if ln.strip()]
    with open(feats) as f:
        h = [ln.strip() for ln in f if ln.strip()]
    return w, h


def main(argv=None) -> int:
    """parse the stage's command line and run it with the training contractions in `--precision` (scoped to this call:
    the C ABI holds no arithmetic mode, runtime.train_precision is host-side state restored on return)."""
    from shallow_wavenet_amd.runtime import train_precision
    args = build_parser().parse_args(argv)
    with train_precision(args.precision):
        return _run(args)


def _run(args) -> int:
    if args.GPU_device is not None:
        os.environ["HIP_VISIBLE_DEVICES"] = str(args.GPU_device)
    os.makedirs(args.expdir, exist_ok=True)
    logging.basicConfig(level=logging.INFO if args.verbose >= 1 else logging.WARN,
                        format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s",
                        datefmt="%m/%d/%Y %I:%M:%S", filename=os.path.join(args.expdir, "train.log"))
    logging.getLogger().addHandler(logging.StreamHandler())
    os.environ["PYTHONHASHSEED"] = str(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    artefacts.save_config(args, os.path.join(args.expdir, "model.conf"))      # the Namespace, train_cswnv...py:293
    if not torch.cuda.is_available():
        logging.error("gpu is not available. please check the setting.")
        return 1
    if args.precision != "fp32":
        logging.info("training contractions in %s operands, fp32 accumulation" % args.precision)
    from shallow_wavenet_amd.nets.cswnv_shift1 import CSWNV, LaplaceLoss, LSDloss, initialize
    model = CSWNV(n_aux=args.n_aux, skip_chn=args.skip_chn, hid_chn=args.hid_chn, dilation_depth=args.dilation_depth,
                  dilation_repeat=args.dilation_repeat, kernel_size=args.kernel_size,
                  aux_kernel_size=args.aux_kernel_size, aux_dilation_size=args.aux_dilation_size,
                  do_prob=args.do_prob, seg=args.seg, lpc=args.lpc, aux_conv2d_flag=args.aux_conv2d_flag,
                  wav_conv_flag=args.wav_conv_flag, upsampling_factor=args.upsampling_factor)
    model.dropout_source = args.dropout_source
    dev_rng = args.dropout_source == "device"          # the loss's reparameterisation noise follows the same choice
    logging.info(model)
    criterion_lsd, criterion_laplace = LSDloss().cuda(), LaplaceLoss().cuda()
    dev = torch.device("cuda")
    loader = None
    if args.synthetic > 0:
        wav_list, feat_list, loader = synthetic_corpus(args.synthetic, args.n_aux, args.upsampling_factor, seed=args.seed)
        n_eval = max(1, args.synthetic // 8)
        wav_eval, feat_eval = wav_list[:n_eval], feat_list[:n_eval]
        mean, scale = np.zeros(args.n_aux), np.ones(args.n_aux)
    else:
        wav_list, feat_list = _file_lists(args.waveforms, args.feats)
        wav_eval, feat_eval = _file_lists(args.waveforms_eval, args.feats_eval)
        mean, scale = read_stats(args.stats, args.string_path)
    assert len(wav_list) == len(feat_list) and len(wav_eval) == len(feat_eval)
    model.cuda()
    model.train()
    model.apply(initialize)
    set_scale_in(model, mean, scale)
    n_train = sum(int(np.prod(p.size())) for p in model.parameters() if p.requires_grad) / 1e6
    logging.info("Trainable Parameters: %.3f million" % n_train)
    optimizer = make_adam(optimizer_parameters(model), args.lr)
    epoch_idx = 0
    checkpoint = None
    if args.pretrained is not None:
        checkpoint = artefacts.load_checkpoint(args.pretrained)
        model.load_state_dict(checkpoint["model"])
        logging.info("pretrained from %d-iter checkpoint." % checkpoint["iterations"])
    elif args.resume is not None:
        checkpoint = artefacts.load_checkpoint(args.resume)
        model.load_state_dict(checkpoint["model"])
        optimizer.load_state_dict(checkpoint["optimizer"])
        epoch_idx = checkpoint["iterations"]
        logging.info("restored from %d-iter checkpoint." % epoch_idx)
    logging.info("number of training data = %d." % len(wav_list))
    logging.info("number of evaluation data = %d." % len(wav_eval))
    gen = train_generator(wav_list, feat_list, model.receptive_field, args.string_path, args.batch_size, model.seg,
                          True, args.upsampling_factor, dev, loader)
    gen_eval = train_generator(wav_eval, feat_eval, model.receptive_field, args.string_path, args.batch_size,
                               model.seg, False, args.upsampling_factor, dev, loader)
    fft_facts = fft_sizes(args.n_fft_facts)
    hann_win = [torch.hann_window(n).cuda() for n in fft_facts]
    logging.info(fft_facts)
    if args.resume is not None:
        np.random.set_state(checkpoint["numpy_random_state"])
        torch.set_rng_state(checkpoint["torch_random_state"])
    loss_laplace, loss_err, loss_lsd = [], [], []
    total, iter_count, iter_idx = 0.0, 0, 0
    min_eval = (float("inf"),) * 6
    min_idx = -1
    logging.info("==%d EPOCH==" % (epoch_idx + 1))
    logging.info("Training data")
    while epoch_idx < args.epoch_count:
        start = time.time()
        x, h, c_idx, utt_idx, wavfile, h_bs, x_bs, h_ss, x_ss = next(gen)
        if c_idx < 0:   # ---- end of epoch: checkpoint, report, evaluate (train_cswnv...py:497-680)
            numpy_random_state, torch_random_state = np.random.get_state(), torch.get_rng_state()
            save_checkpoint(args.expdir, model, optimizer, numpy_random_state, torch_random_state, epoch_idx + 1)
            mlsd, slsd = (np.mean(loss_lsd), np.std(loss_lsd)) if loss_lsd else (float("nan"), float("nan"))
            logging.info("(EPOCH:%d) average training loss = %.6f (+- %.6f) %.6f dB (+- %.6f dB) %.6f "
                         "(+- %.6f) (%.3f min., %.3f sec / batch)" % (
                             epoch_idx + 1, np.mean(loss_laplace), np.std(loss_laplace), mlsd, slsd,
                             np.mean(loss_err), np.std(loss_err), total / 60.0, total / max(iter_count, 1)))
            ev_lap, ev_lsd, ev_err = [], [], []
            model.eval()
            etotal, ecount = 0.0, 0
            logging.info("Evaluation data")
            while True:
                estart = time.time()
                x, h, c_idx, utt_idx, wavfile, h_bs, x_bs, h_ss, x_ss = next(gen_eval)
                if c_idx < 0:
                    break
                with torch.no_grad():
                    bh, bx, trg, xp, flen = slice_chunk(model, x, h, h_bs, x_bs, h_ss, x_ss)
                    _, l_lap, l_lsd, l_err = batch_loss(model, criterion_laplace, criterion_lsd, bh, bx, trg, xp, flen,
                                                        h_ss, fft_facts, hann_win, do=False, eps_on_device=dev_rng)
                ev_lap.append(l_lap.item()); ev_err.append(l_err.item())
                if l_lsd is not None:
                    ev_lsd.append(l_lsd.item())
                etotal += time.time() - estart
                ecount += 1
            elsd, eslsd = (np.mean(ev_lsd), np.std(ev_lsd)) if ev_lsd else (float("nan"), float("nan"))
            logging.info("(EPOCH:%d) average evaluation loss = %.6f (+- %.6f) %.6f dB (+- %.6f dB) %.6f "
                         "(+- %.6f) (%.3f min., %.3f sec / batch)" % (
                             epoch_idx + 1, np.mean(ev_lap), np.std(ev_lap), elsd, eslsd, np.mean(ev_err),
                             np.std(ev_err), etotal / 60.0, etotal / max(ecount, 1)))
            cur = (np.mean(ev_lap), np.std(ev_lap), elsd, eslsd, np.mean(ev_err), np.std(ev_err))
            if not np.isfinite(sum(min_eval)) or sum(cur) <= sum(min_eval):       # sum of the six figures (:645-654)
                min_eval, min_idx = cur, epoch_idx
            logging.info("min_eval_loss = %.6f (+- %.6f) %.6f dB (+- %.6f dB) %.6f (+- %.6f) min_idx=%d" % (
                *min_eval, min_idx + 1))
            loss_laplace, loss_err, loss_lsd = [], [], []
            total, iter_count = 0.0, 0
            epoch_idx += 1
            np.random.set_state(numpy_random_state)          # evaluation must not advance the training streams (:664-665)
            torch.set_rng_state(torch_random_state)
            model.train()
            for p in model.parameters():
                p.requires_grad = True
            for p in model.scale_in.parameters():
                p.requires_grad = False
            if epoch_idx < args.epoch_count:
                logging.info("==%d EPOCH==" % (epoch_idx + 1))
                logging.info("Training data")
            continue
        # ---- one training chunk (train_cswnv...py:700-874)
        logging.info("%d iteration [%d]" % (iter_idx + 1, epoch_idx + 1))
        tf, ts = h.shape[0], x.shape[0]
        bh, bx, trg, xp, flen = slice_chunk(model, x, h, h_bs, x_bs, h_ss, x_ss)
        loss, l_lap, l_lsd, l_err = batch_loss(model, criterion_laplace, criterion_lsd, bh, bx, trg, xp, flen, h_ss,
                                               fft_facts, hann_win, do=True, eps_on_device=dev_rng)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        loss_err.append(l_err.item()); loss_laplace.append(l_lap.item())
        tag = os.path.basename(os.path.dirname(wavfile)) + "/" + os.path.basename(wavfile)
        if l_lsd is not None:
            loss_lsd.append(l_lsd.item())
            logging.info("batch loss %s [%d:%d] %d %d %d %d %d %d = %.3f %.3f dB %.6f (%.3f sec)" % (
                tag, c_idx + 1, utt_idx + 1, tf, ts, h_ss, h_bs, x_ss, x_bs, l_lap.item(), l_lsd.item(), l_err.item(),
                time.time() - start))
        else:
            logging.info("batch loss %s [%d:%d] %d %d %d %d %d %d = %.3f n/a %.6f (%.3f sec)" % (
                tag, c_idx + 1, utt_idx + 1, tf, ts, h_ss, h_bs, x_ss, x_bs, l_lap.item(), l_err.item(),
                time.time() - start))
        iter_idx += 1
        iter_count += 1
        total += time.time() - start
        if args.max_iters and iter_idx >= args.max_iters:
            break
    torch.save({"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}},
               os.path.join(args.expdir, "checkpoint-final.pkl"))
    logging.info("final checkpoint created.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
