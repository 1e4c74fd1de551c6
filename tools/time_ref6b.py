import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tools"))
from shallow_wavenet_amd import config as C
import time_decode as T
Tf = int(sys.argv[1]) if len(sys.argv) > 1 else 40
T.run(C.ref6_laplace(1, 4), 1, Tf, variants=(3,), reps=2)
T.run(C.ref6_laplace(5, 4), 1, Tf, variants=(3,), reps=1)
T.run(C.ref6_softmax(), 1, Tf, variants=(3,), reps=1)
T.run(C.ref6_laplace(1, 4), 8, Tf, variants=(3,), reps=1)
T.run(C.ref6_laplace(1, 4), 32, Tf, variants=(3,), reps=1)
