#!/usr/bin/env python3
"""Stage-4 entry point with the reference's script name and flags (run.sh:588-615):
`python train_cswnv_laplace-stftcmplx_shift1.py --waveforms ... --feats ... --stats ... --expdir ...`
(`--synthetic N` trains on generated utterances).  The work is in shallow_wavenet_amd/train_driver.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from shallow_wavenet_amd.train_driver import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
