#!/usr/bin/env python3
"""stage 2 of run.sh - counterpart of src/bin/calc_stats.py: same flags; see shallow_wavenet_amd/featio.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from shallow_wavenet_amd.featio import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
