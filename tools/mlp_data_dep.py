"""GPU box tool: is mlp_fused's duration data-dependent?  The production kernel (no stamps) on random rows, on identical rows and on
zero rows, N launches each; read the durations with rocprofv3 --kernel-trace (launch order: random, same, zero)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tempfile
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=False)
eng = Engine(d, precision="bf16")
M = 1280 * 128
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
g = np.ones(384, np.float32); b = np.zeros(384, np.float32)
w1 = (rng.standard_normal((1536, 384)) / 20).astype(np.float32); b1 = np.zeros(1536, np.float32)
w2 = (rng.standard_normal((384, 1536)) / 40).astype(np.float32); b2 = np.zeros(384, np.float32)
wp = (rng.standard_normal((384, 384)) / 20).astype(np.float32)
xr = rng.standard_normal((M, 384)).astype(np.float32); ar = rng.standard_normal((M, 384)).astype(np.float32)
xs = np.broadcast_to(xr[:1], (M, 384)).copy(); as_ = np.broadcast_to(ar[:1], (M, 384)).copy()
xz = np.zeros((M, 384), np.float32)
for name, x, att in (("random", xr, ar), ("same", xs, as_), ("zero", xz, xz)):
    for _ in range(n):
        eng.dbg_mlp(x, g, b, w1, b1, w2, b2, g, b, att=att, wp=wp, bp=b)
    print(name, "done", flush=True)
