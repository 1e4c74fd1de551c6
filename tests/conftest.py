import ast
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# the GPU box exposes every host core to os.cpu_count() but grants ~16: an oversubscribed
# torch thread pool makes the CPU oracle crawl
try:
    import torch as _torch
    _torch.set_num_threads(max(1, min(8, len(os.sched_getaffinity(0)))))
except Exception:  # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_golden(name):
    """-> (NetConfig, dict of arrays) ; fixtures never contain pickles."""
    from shallow_wavenet_amd.config import NetConfig
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    cfg = NetConfig(**ast.literal_eval(str(d["cfg_json"]))) if "cfg_json" in d else None
    return cfg, d


@pytest.fixture(scope="session")
def gpu_ok():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return True
