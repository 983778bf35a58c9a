"""GPU box tool: panel stamps of mlp_pair (workgroup 0, wave 0, first panels).  python tools/pair_stamps.py [crops] [ablate]"""
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=False)
eng = Engine(d, precision="bf16")
assert eng.set_tuning(b"dec_stamps", 5) == 0 and eng.set_tuning(b"mlp_pair", 1) == 0
if len(sys.argv) > 2:
    assert eng.set_tuning(b"pair_ablate", int(sys.argv[2])) == 0
M = (int(sys.argv[1]) if len(sys.argv) > 1 else 1280) * 128
rng = np.random.default_rng(0)
x = rng.standard_normal((M, 384)).astype(np.float32)
g = np.ones(384, np.float32); b = np.zeros(384, np.float32)
w1 = (rng.standard_normal((1536, 384)) / 20).astype(np.float32); b1 = np.zeros(1536, np.float32)
w2 = (rng.standard_normal((384, 1536)) / 40).astype(np.float32); b2 = np.zeros(384, np.float32)
eng.dbg_mlp(x, g, b, w1, b1, w2, b2, g, b)
buf = (C.c_ulonglong * (26 * 16))()
assert eng.lib.ttr_dbg_dec_stamps(buf) == 0
tp = np.array(buf[384:400], dtype=np.uint64).reshape(4, 4).astype(np.float64)
for k in range(4):
    a = tp[k]
    if a[3] > a[0] > 0:
        print(f"panel {k}: front {a[1]-a[0]:.0f}  loop {a[2]-a[1]:.0f} ({(a[2]-a[1])/49:.0f} per iteration)  epilogue {a[3]-a[2]:.0f}  cycles" + (f"  gap to next {tp[k+1][0]-a[3]:.0f}" if k < 3 and tp[k+1][0] > 0 else ""))
