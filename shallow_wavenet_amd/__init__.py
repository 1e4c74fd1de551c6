"""MI355X-native hot path of the shallow WaveNet vocoder (decode loop + teacher-forced stack).

Host side mirrors the reference's `src/nets` API (`nets/cswnv_shift1.py`, `nets/dswnv.py`);
all arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of
`include/swn_hip.h` (`csrc/`).  There is no CPU fallback: product paths raise if the HIP
library is missing.
"""
from .config import NetConfig  # noqa: F401

__all__ = ["NetConfig"]
