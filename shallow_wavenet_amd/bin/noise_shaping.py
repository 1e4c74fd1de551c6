#!/usr/bin/env python3
"""stages 3 / 6 / 9 of run.sh - counterpart of src/bin/noise_shaping.py: same flags; see shallow_wavenet_amd/noise_shaping_driver.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from shallow_wavenet_amd.noise_shaping_driver import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
