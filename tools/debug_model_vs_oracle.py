import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import weights as W
from asr_amd.model import DeeplabModel
from oracle.model import OracleDeeplabV3Plus
w = W.make_synthetic_weights(1234)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = int(sys.argv[2]) if len(sys.argv) > 2 else 4
x = np.random.default_rng(0).random((b, size, size, 3), dtype=np.float32)
ref = OracleDeeplabV3Plus(w).forward(x)
m = DeeplabModel(w, (size, size, 3), 21, False, None, precision=os.environ.get('ASR_PRECISION'))
print('precision', m.precision)
for rep in range(3):
    got = m.predict(x, batch_size=b)
    print(size, b, "rep", rep, "maxdiff", np.abs(got - ref).max(), "scale", np.abs(ref).max(), "argmax agree", (got.argmax(-1) == ref.argmax(-1)).mean(), flush=True)
