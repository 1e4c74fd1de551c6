#!/usr/bin/env python3
"""stage 5/8 driver, Laplace model - counterpart of src/bin/decode_cswnv_laplace-shift1.py
(run.sh:675-684): same flags, same outputs; see shallow_wavenet_amd/decode_driver.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from shallow_wavenet_amd.decode_driver import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main("laplace"))
