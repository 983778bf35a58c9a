"""Debug helper (GPU box): run the full-size CCL parity case several times in one process and print any difference."""
import sys, os, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import post
from tests.golden.make_golden import synthetic_heatmap
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine

d = tempfile.mkdtemp()
W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d, precision="f32")
big = synthetic_heatmap(7, 512, 384)
ref, _, _ = post.get_detected_boxes(big[..., 0], big[..., 1])
for it in range(6):
    got = eng.ccl_boxes(big)
    if got.shape != ref.shape:
        print(it, "shape", got.shape, ref.shape)
        continue
    bad = np.nonzero((got != ref).any(1))[0]
    print(it, "mismatching rows:", bad.tolist())
    for b in bad[:6]:
        print("   got", got[b].tolist(), "\n   ref", ref[b].tolist())
