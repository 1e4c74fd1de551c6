"""Stage-7 training driver for the softmax model: the host logic of the reference's `src/bin/train_dswnv_softmax.py`
around the HIP forward / backward of `nets/dswnv.DSWNV` (row f2 of SURVEY.md section 8).

Same structure as `train_driver.py` (file readers, synthetic corpus, checkpoint dictionary, scale_in set-up are
shared); what differs and is restated here with the reference's line numbers: the chunk plan
(train_dswnv_softmax.py:96-146), the mu-law class targets, the cross-entropy loss on the logits past the receptive
field (:549-575) and the one-figure log lines (:470-575).  The drop-in module takes class indices, so the one-hot
tensor the reference builds per utterance (`OneHot`, :123-131) is never materialised."""
from __future__ import annotations

import argparse
import logging
import os
import sys
import time
from typing import List, Tuple

import numpy as np
import torch

from . import artefacts
from .train_driver import (_file_lists, make_adam, read_feat, read_stats, read_wav, save_checkpoint, set_scale_in,
                           synthetic_corpus, validate_length)


def chunk_plan(n_frames: int, receptive_field: int, batch_size: int, upsampling_factor: int) -> List[Tuple[int, int, int, int]]:
    """(h_bs, x_bs, h_ss, x_ss) per chunk (train_dswnv_softmax.py:96-146): chunks of (rf + batch_size + 1) // U frames
    while MORE than that many frames are left, then one open-ended tail chunk if more than rf + 1 samples remain."""
    U = upsampling_factor
    bs = batch_size
    if bs != 0 and U != 0:
        bs -= (receptive_field + bs + 1) % U
    h_bs = (receptive_field + bs + 1) // U
    x_bs = h_bs * U
    delta = bs // U
    out, len_frm, h_ss, x_ss = [], n_frames, 0, 0
    while True:
        if len_frm > h_bs:
            out.append((h_bs, x_bs, h_ss, x_ss))
            h_ss += delta
            x_ss = h_ss * U
            len_frm -= delta
        elif len_frm * U > receptive_field + 1:
            out.append((-1, -1, h_ss, x_ss))
            break
        else:
            break
    return out


def train_generator(wav_list, feat_list, receptive_field, string_path, batch_size, n_quantize, training,
                    upsampling_factor, device=None, loader=None):
    """yields (x_class, h, c_idx, utt_idx, wavfile, h_bs, x_bs, h_ss, x_ss); c_idx = -1 closes an epoch."""
    from .nets.dswnv import encode_mu_law
    n_files = len(wav_list)
    idx = np.random.permutation(n_files) if training else np.arange(n_files)
    while True:
        for c_idx, i in enumerate(idx):
            wavfile, featfile = wav_list[i], feat_list[i]
            x, h = loader(wavfile, featfile) if loader is not None else (read_wav(wavfile), read_feat(featfile, string_path))
            x, h = validate_length(np.asarray(x, dtype=np.float32), np.asarray(h), upsampling_factor)
            xc = torch.as_tensor(encode_mu_law(x, n_quantize), dtype=torch.int64, device=device)
            ht = torch.as_tensor(h, dtype=torch.float32, device=device)
            for h_bs, x_bs, h_ss, x_ss in chunk_plan(len(h), receptive_field, batch_size, upsampling_factor):
                yield xc, ht, c_idx, int(i), wavfile, h_bs, x_bs, h_ss, x_ss
        yield [], [], -1, -1, [], [], [], [], []
        if training:
            idx = np.random.permutation(n_files)


def slice_chunk(x_class: torch.Tensor, h: torch.Tensor, h_bs: int, x_bs: int, h_ss: int, x_ss: int):
    """-> batch_h (1, n_aux, Tf'), input classes (1, T'), target classes (T')   (train_dswnv_softmax.py:549-561)."""
    bh, xc = h[h_ss:], x_class[x_ss:]
    if h_bs != -1:
        bh, target, inp = bh[:h_bs], xc[1:x_bs], xc[:x_bs - 1]
    else:
        target, inp = xc[1:], xc[:-1]
    return bh.transpose(0, 1).unsqueeze(0), inp.unsqueeze(0), target


def batch_loss(model, criterion, batch_h, batch_x, target, h_ss: int, do: bool):
    """cross entropy of the logits against the next-sample classes, past the receptive field for every chunk but
    the first (train_dswnv_softmax.py:563-569)."""
    out = model(batch_x, batch_h, do=do)[0]
    rf = model.receptive_field
    return criterion(out[rf:], target[rf:]) if h_ss > 0 else criterion(out, target)


def optimizer_parameters(model):
    """train_dswnv_softmax.py:352-359"""
    mods = [model.conv_aux, model.upsampling]
    if model.wav_conv_flag:
        mods.append(model.wav_conv)
    mods += [model.causal, model.in_x, model.dil_h, model.out_skip, model.out_1, model.out_2]
    return [p for m in mods for p in m.parameters()]


def build_parser() -> argparse.ArgumentParser:
    """flags of train_dswnv_softmax.py:185-250 (+ --synthetic / --max_iters)"""
    sb = lambda v: str(v).lower() in ("1", "true", "yes", "y", "t", "on")
    p = argparse.ArgumentParser()
    for k in ("waveforms", "waveforms_eval", "feats", "feats_eval", "stats"):
        p.add_argument("--" + k, type=str)
    p.add_argument("--expdir", required=True, type=str)
    p.add_argument("--n_quantize", default=256, type=int)
    p.add_argument("--n_aux", default=39, type=int)
    p.add_argument("--dilation_depth", default=3, type=int)
    p.add_argument("--dilation_repeat", default=3, type=int)
    p.add_argument("--hid_chn", default=192, type=int)
    p.add_argument("--skip_chn", default=256, type=int)
    p.add_argument("--kernel_size", default=6, type=int)
    p.add_argument("--aux_kernel_size", default=3, type=int)
    p.add_argument("--aux_dilation_size", default=2, type=int)
    p.add_argument("--upsampling_factor", default=110, type=int)
    p.add_argument("--string_path", default="/feat_org_lf0", type=str)
    p.add_argument("--lr", default=1e-4, type=float)
    p.add_argument("--batch_size", default=1100, type=int)
    p.add_argument("--epoch_count", default=500, type=int)
    p.add_argument("--do_prob", default=0, type=float)
    p.add_argument("--wav_conv_flag", default=False, type=sb)
    p.add_argument("--audio_in", default=False, type=sb)
    p.add_argument("--seed", default=1, type=int)
    p.add_argument("--resume", default=None, type=str)
    p.add_argument("--pretrained", default=None, type=str)
    p.add_argument("--GPU_device", default=None, type=int)
    p.add_argument("--verbose", default=1, type=int)
    p.add_argument("--synthetic", default=0, type=int)
    p.add_argument("--max_iters", default=0, type=int)
    p.add_argument("--dropout_source", default="device", choices=["device", "host"],
                   help="where the nn.Dropout masks are drawn: the model's device (like the reference on a GPU) or the torch "
                        "CPU generator in the reference's order (reproduces a CPU run of the reference; ~1 s per chunk)")
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                   help="arithmetic of the training step's contractions (not a reference flag): fp32 = parity mode, "
                        "bf16 = bf16 operands with fp32 accumulation (SWN_PRECISION_BF16)")
    return p


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
    artefacts.save_config(args, os.path.join(args.expdir, "model.conf"))      # the Namespace, train_dswnv_softmax.py
    if not torch.cuda.is_available():
        logging.error("gpu is not available. please check the setting.")
        return 1
    if args.precision != "fp32":
        logging.info("training contractions in %s operands, fp32 accumulation" % args.precision)
    from .nets.dswnv import DSWNV, initialize
    model = DSWNV(n_quantize=args.n_quantize, n_aux=args.n_aux, hid_chn=args.hid_chn, skip_chn=args.skip_chn,
                  dilation_depth=args.dilation_depth, dilation_repeat=args.dilation_repeat, kernel_size=args.kernel_size,
                  aux_kernel_size=args.aux_kernel_size, aux_dilation_size=args.aux_dilation_size,
                  audio_in_flag=args.audio_in, do_prob=args.do_prob, wav_conv_flag=args.wav_conv_flag,
                  upsampling_factor=args.upsampling_factor)
    model.dropout_source = args.dropout_source
    logging.info(model)
    criterion = torch.nn.CrossEntropyLoss().cuda()
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
    model.cuda()
    model.train()
    model.apply(initialize)
    set_scale_in(model, mean, scale)
    logging.info("Trainable Parameters: %.3f million" % (sum(int(np.prod(p.size())) for p in model.parameters() if p.requires_grad) / 1e6))
    optimizer = make_adam(optimizer_parameters(model), args.lr)
    epoch_idx, checkpoint = 0, None
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
    mk = lambda w, f, tr: train_generator(w, f, model.receptive_field, args.string_path, args.batch_size, args.n_quantize,
                                          tr, args.upsampling_factor, dev, loader)
    gen, gen_eval = mk(wav_list, feat_list, True), mk(wav_eval, feat_eval, False)
    if args.resume is not None:
        np.random.set_state(checkpoint["numpy_random_state"])
        torch.set_rng_state(checkpoint["torch_random_state"])
    loss, total, iter_idx, iter_count = [], 0.0, 0, 0
    min_eval_loss, min_eval_loss_std, min_idx = 99999999.99, 99999999.99, -1
    logging.info("==%d EPOCH==" % (epoch_idx + 1))
    logging.info("Training data")
    while epoch_idx < args.epoch_count:
        start = time.time()
        xc, h, c_idx, utt_idx, wavfile, h_bs, x_bs, h_ss, x_ss = next(gen)
        if c_idx < 0:
            numpy_random_state, torch_random_state = np.random.get_state(), torch.get_rng_state()
            save_checkpoint(args.expdir, model, optimizer, numpy_random_state, torch_random_state, epoch_idx + 1)
            logging.info("(EPOCH:%d) average training loss = %.6f (+- %.6f) (%.3f min., %.3f sec / batch)" % (
                epoch_idx + 1, np.mean(np.array(loss, dtype=np.float64)), np.std(np.array(loss, dtype=np.float64)),
                total / 60.0, total / max(iter_count, 1)))
            ev, etotal, ecount = [], 0.0, 0
            model.eval()
            logging.info("Evaluation data")
            with torch.no_grad():
                while True:
                    estart = time.time()
                    xc, h, c_idx, utt_idx, wavfile, h_bs, x_bs, h_ss, x_ss = next(gen_eval)
                    if c_idx < 0:
                        break
                    bh, bx, trg = slice_chunk(xc, h, h_bs, x_bs, h_ss, x_ss)
                    l = batch_loss(model, criterion, bh, bx, trg, h_ss, do=False)
                    ev.append(l.item())
                    logging.info("batch eval loss %s [%d:%d] %d %d %d %d %d %d = %.3f (%.3f sec)" % (
                        os.path.basename(os.path.dirname(wavfile)) + "/" + os.path.basename(wavfile), c_idx + 1, utt_idx + 1,
                        h.shape[0], xc.shape[0], h_ss, h_bs, x_ss, x_bs, l.item(), time.time() - estart))
                    etotal += time.time() - estart
                    ecount += 1
            eval_loss, eval_loss_std = np.mean(np.array(ev, dtype=np.float64)), np.std(np.array(ev, dtype=np.float64))
            logging.info("(EPOCH:%d) average evaluation loss = %.6f (+- %.6f) (%.3f min., %.3f sec / batch)" % (
                epoch_idx + 1, eval_loss, eval_loss_std, etotal / 60.0, etotal / max(ecount, 1)))
            if eval_loss + eval_loss_std <= min_eval_loss + min_eval_loss_std:
                min_eval_loss, min_eval_loss_std, min_idx = eval_loss, eval_loss_std, epoch_idx
            logging.info("min_eval_loss=%.6f (+- %.6f), min_idx=%d" % (min_eval_loss, min_eval_loss_std, min_idx + 1))
            loss, total, iter_count = [], 0.0, 0
            epoch_idx += 1
            np.random.set_state(numpy_random_state)
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
        logging.info("%d iteration [%d]" % (iter_idx + 1, epoch_idx + 1))
        bh, bx, trg = slice_chunk(xc, h, h_bs, x_bs, h_ss, x_ss)
        l = batch_loss(model, criterion, bh, bx, trg, h_ss, do=True)
        optimizer.zero_grad()
        l.backward()
        optimizer.step()
        loss.append(l.item())
        logging.info("batch loss %s [%d:%d] %d %d %d %d %d %d = %.3f (%.3f sec)" % (
            os.path.basename(os.path.dirname(wavfile)) + "/" + os.path.basename(wavfile), c_idx + 1, utt_idx + 1,
            h.shape[0], xc.shape[0], h_ss, h_bs, x_ss, x_bs, l.item(), time.time() - start))
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
