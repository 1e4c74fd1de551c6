import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tools"))
from shallow_wavenet_amd import config as C
import time_decode as T
T.run(C.ref6_laplace(1, 4), 64, 10, variants=(4,), reps=1)
