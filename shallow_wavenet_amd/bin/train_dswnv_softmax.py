#!/usr/bin/env python3
"""Stage-7 entry point with the reference's script name and flags (run.sh:792-803); the work is in
shallow_wavenet_amd/train_softmax_driver.py (`--synthetic N` trains on generated utterances)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from shallow_wavenet_amd.train_softmax_driver import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
