"""dev tool: REF6 decode, stepped (3) vs cohort (4), steady-state step cost for several batch sizes."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tools"))
from shallow_wavenet_amd import config as C
from try_cluster import timing
for B in (8, 16, 32, 64):
    timing(C.ref6_laplace(1, 4), B, 4, variants=(3, 4))
timing(C.ref6_laplace(5, 4), 64, 4, variants=(3, 4))
timing(C.ref6_softmax(), 64, 4, variants=(3, 4))
