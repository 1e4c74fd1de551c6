import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tools"))
from shallow_wavenet_amd import config as C
import time_decode as T
Tf = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for B in (8, 64):
    T.run(C.ref6_laplace(1, 4), B, Tf, variants=(3, 4), reps=1)
T.run(C.ref6_softmax(), 64, Tf, variants=(4,), reps=1)
