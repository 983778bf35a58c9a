"""GPU box tool for rocprofv3 passes over the recogniser alone: PARSeq on one batch of random crops (default 1280 = the benchmark's
32 pages x 40 crops), a few forwards, nothing else.   python3 tools/prof_parseq.py [crops] [iterations] [key=value ...]"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
it = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = tempfile.mkdtemp()
W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d, precision=os.environ.get("TTR_PREC", "bf16"))
const = "const" in sys.argv[3:]                # all crops one flat grey: near-constant MFMA operands (the DVFS / power comparison)
for kv in sys.argv[3:]:                      # key=value tuning knobs (Engine.set_tuning)
    if kv == "const":
        continue
    k, v = kv.split("=")
    assert eng.set_tuning(k, int(v)) == 0, kv
crops = np.random.default_rng(0).integers(0, 256, (N, 32, 128, 3), dtype=np.uint8)
if const:
    crops[:] = 128
for _ in range(it):
    lg, ids = eng.parseq_logits(crops)
print("crops", N, "iterations", it, "finite", bool(np.isfinite(lg).all()))
