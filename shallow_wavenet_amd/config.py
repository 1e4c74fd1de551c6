"""Network geometry of the shallow WaveNet vocoders (hot path only).

One immutable description shared by the host mirror (`nets/`), the weight packer,
the HIP launchers, the oracle and the tests.  Everything here is derived from the
constructor arguments of the reference modules:

  * CSWNV.__init__  /root/reference/src/nets/cswnv_shift1.py:130-189
  * DSWNV.__init__  /root/reference/src/nets/dswnv.py:190-248

Reference quirks reproduced on purpose:
  * dilation of stack layer l is K**(l mod dd), not 2**l   (cswnv_shift1.py:77-78)
  * padding[l] = K**(d+1) - K**d, rf = sum(padding) + K - 1 (cswnv_shift1.py:170-183)
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class NetConfig:
    kind: str = "laplace"          # "laplace" (CSWNV) | "softmax" (DSWNV)
    n_aux: int = 54
    hid_chn: int = 192
    skip_chn: int = 256
    aux_kernel_size: int = 3
    aux_dilation_size: int = 2
    dilation_depth: int = 3
    dilation_repeat: int = 2
    kernel_size: int = 7
    upsampling_factor: int = 110
    seg: int = 1                   # laplace only; softmax is always 1
    lpc: int = 0                   # laplace only
    n_quantize: int = 256          # softmax only
    wav_conv_flag: bool = False
    audio_in_flag: bool = False    # softmax only
    aux_conv2d_flag: bool = False  # laplace only

    # ---- derived geometry -------------------------------------------------
    @property
    def H(self) -> int:
        return self.hid_chn

    @property
    def S(self) -> int:
        return self.skip_chn

    @property
    def K(self) -> int:
        return self.kernel_size

    @property
    def U(self) -> int:
        return self.upsampling_factor

    @property
    def L(self) -> int:
        return self.dilation_depth * self.dilation_repeat

    @property
    def dil_facts(self) -> List[int]:
        return [d for d in range(self.dilation_depth)] * self.dilation_repeat

    @property
    def dilations(self) -> List[int]:
        return [self.kernel_size ** d for d in self.dil_facts]

    @property
    def paddings(self) -> List[int]:
        return [self.kernel_size ** (d + 1) - self.kernel_size ** d for d in self.dil_facts]

    @property
    def receptive_field(self) -> int:
        return sum(self.paddings) + self.kernel_size - 1

    @property
    def A0(self) -> int:
        """conditioning channels after conv_aux (n_aux * k**layers)."""
        return self.n_aux * self.aux_kernel_size ** self.aux_dilation_size

    @property
    def A(self) -> int:
        """input channels of in_x (cswnv_shift1.py:158-162, dswnv.py:214-221)."""
        if self.kind == "softmax":
            return self.A0 + (self.n_quantize if self.audio_in_flag else 0)
        if self.seg > 1 and not self.aux_conv2d_flag:
            return self.A0 * self.seg
        return self.A0

    @property
    def causal_in(self) -> int:
        """input channels of `causal` (cswnv_shift1.py:163-167, dswnv.py:222-226)."""
        if self.wav_conv_flag:
            return self.hid_chn
        return 1 if self.kind == "laplace" else self.n_quantize

    @property
    def n_out(self) -> int:
        """width of out_2 (cswnv_shift1.py:189, dswnv.py:248)."""
        if self.kind == "softmax":
            return self.n_quantize
        return 2 * self.seg + self.lpc

    @property
    def out1_chn(self) -> int:
        """width of out_1 (cswnv_shift1.py:188 S->S, dswnv.py:247 S->Q)."""
        return self.n_quantize if self.kind == "softmax" else self.skip_chn

    def ctor_kwargs(self) -> Dict:
        """kwargs of the reference-compatible constructor for this geometry."""
        common = dict(n_aux=self.n_aux, hid_chn=self.hid_chn, skip_chn=self.skip_chn,
                      aux_kernel_size=self.aux_kernel_size,
                      aux_dilation_size=self.aux_dilation_size,
                      dilation_depth=self.dilation_depth,
                      dilation_repeat=self.dilation_repeat,
                      kernel_size=self.kernel_size,
                      upsampling_factor=self.upsampling_factor,
                      wav_conv_flag=self.wav_conv_flag)
        if self.kind == "softmax":
            common.update(n_quantize=self.n_quantize, audio_in_flag=self.audio_in_flag)
        else:
            common.update(seg=self.seg, lpc=self.lpc, aux_conv2d_flag=self.aux_conv2d_flag)
        return common

    def param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """state_dict keys and shapes in reference construction order
        (order = cswnv_shift1.py:151-189, dswnv.py:210-248)."""
        n, k = self.n_aux, self.aux_kernel_size
        out: List[Tuple[str, Tuple[int, ...]]] = []
        out += [("scale_in.weight", (n, n, 1)), ("scale_in.bias", (n,))]
        for i in range(self.aux_dilation_size):
            cin, cout = n * k ** i, n * k ** (i + 1)
            out += [(f"conv_aux.conv.{i}.weight", (cout, cin, k)),
                    (f"conv_aux.conv.{i}.bias", (cout,))]
        out += [("upsampling.conv.weight", (1, 1, 1, self.U)), ("upsampling.conv.bias", (1,))]
        if self.kind == "laplace" and self.aux_conv2d_flag and self.seg > 1:
            out += [("aux_conv2d.weight", (self.A0, self.A0, self.seg, 1)),
                    ("aux_conv2d.bias", (self.A0,))]
        if self.wav_conv_flag:
            wc_in = 1 if self.kind == "laplace" else self.n_quantize
            out += [("wav_conv.weight", (self.H, wc_in, 1)), ("wav_conv.bias", (self.H,))]
        out += [("causal.conv.weight", (self.H, self.causal_in, self.K)),
                ("causal.conv.bias", (self.H,))]
        # state_dict order follows module registration: the three ModuleLists one after another
        for l in range(self.L):
            out += [(f"in_x.{l}.weight", (2 * self.H, self.A, 1)), (f"in_x.{l}.bias", (2 * self.H,))]
        for l in range(self.L):
            out += [(f"dil_h.{l}.conv.weight", (2 * self.H, self.H, self.K)),
                    (f"dil_h.{l}.conv.bias", (2 * self.H,))]
        for l in range(self.L):
            out += [(f"out_skip.{l}.weight", (self.S, self.H, 1)), (f"out_skip.{l}.bias", (self.S,))]
        out += [("out_1.weight", (self.out1_chn, self.S, 1)), ("out_1.bias", (self.out1_chn,)),
                ("out_2.weight", (self.n_out, self.out1_chn, 1)), ("out_2.bias", (self.n_out,))]
        return out

    def n_params(self) -> int:
        tot = 0
        for _, shp in self.param_shapes():
            n = 1
            for s in shp:
                n *= s
            tot += n
        return tot

    def to_dict(self) -> Dict:
        return asdict(self)


# ---- named shapes used by BASELINE.json / SURVEY.md section 8 -------------------
def bl6_laplace(seg: int = 1, lpc: int = 0) -> NetConfig:
    """BASELINE-literal 1x6 stack, 64 hidden / 128 skip, 22.05 kHz (cfg2 / cfg3 / cfg5)."""
    return NetConfig(kind="laplace", n_aux=54, hid_chn=64, skip_chn=128, dilation_depth=6,
                     dilation_repeat=1, kernel_size=2, upsampling_factor=110, seg=seg, lpc=lpc,
                     wav_conv_flag=True)


def bl6_softmax() -> NetConfig:
    """cfg1: softmax mu-law 256, 1x6 stack, 64 hidden, 16 kHz."""
    return NetConfig(kind="softmax", n_aux=53, hid_chn=64, skip_chn=256, dilation_depth=6,
                     dilation_repeat=1, kernel_size=2, upsampling_factor=80, n_quantize=256,
                     wav_conv_flag=False)


def ref6_laplace(seg: int = 1, lpc: int = 4) -> NetConfig:
    """reference-shipped shape (run.sh:165-230): 3x2 layers, K=7, H=192, S=256."""
    return NetConfig(kind="laplace", n_aux=54, hid_chn=192, skip_chn=256, dilation_depth=3,
                     dilation_repeat=2, kernel_size=7, upsampling_factor=110, seg=seg, lpc=lpc,
                     wav_conv_flag=True)


def ref6_softmax() -> NetConfig:
    return NetConfig(kind="softmax", n_aux=54, hid_chn=256, skip_chn=256, dilation_depth=3,
                     dilation_repeat=2, kernel_size=7, upsampling_factor=110, n_quantize=256,
                     wav_conv_flag=False)


def tiny(kind: str = "laplace", seg: int = 1, lpc: int = 0, wav_conv_flag: bool = True,
         audio_in_flag: bool = False, aux_conv2d_flag: bool = False) -> NetConfig:
    """G0 fixture shape: n_aux=10 U=20 H=32 S=48 K=3 dd=3 dr=2 (rf=54)."""
    if kind == "softmax":
        return NetConfig(kind="softmax", n_aux=10, hid_chn=32, skip_chn=48, dilation_depth=3,
                         dilation_repeat=2, kernel_size=3, upsampling_factor=20, n_quantize=256,
                         wav_conv_flag=wav_conv_flag, audio_in_flag=audio_in_flag)
    return NetConfig(kind="laplace", n_aux=10, hid_chn=32, skip_chn=48, dilation_depth=3,
                     dilation_repeat=2, kernel_size=3, upsampling_factor=20, seg=seg, lpc=lpc,
                     wav_conv_flag=wav_conv_flag, aux_conv2d_flag=aux_conv2d_flag)
