"""GPU box: PARSeq logits of the f16x4 engine against the CPU oracle with tuning knobs on:  python tools/x4_knob_check.py n_crops key=value ...
(development tool; uses the oracle)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tuatara_amd import weights as W
from tuatara_amd.build import build_lib
from tuatara_amd.engine import Engine
from oracle import pipeline
from tests import parity_rules as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
build_lib()
wdir = "/tmp/x4_knob_w"
c, p = W.make_synthetic_weights(wdir, seed=0, structured=True)
_, parseq = pipeline.load_models(c, p)
e = Engine(wdir, precision="f16x4")
for seed in (0, 1, 2):
    crops = np.random.default_rng(seed).integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
    refs = []
    with torch.no_grad():
        for i in range(0, n, 64):
            refs.append(parseq(torch.from_numpy(crops[i:i + 64]).permute(0, 3, 1, 2).float().div(255.0)).numpy())
    ref = np.concatenate(refs)
    up = R.upto_eos(ref.argmax(-1)); mask = np.arange(26)[None, :] < up[:, None]
    for knobs in ([], sys.argv[2:]):
        for kv in sys.argv[2:]:
            k, v = kv.split("=")
            assert e.set_tuning(k, int(v) if knobs else 0) == 0
        got, ids = e.parseq_logits(crops)
        t = time.time()
        for _ in range(3): e.parseq_logits(crops)
        dt = (time.time() - t) / 3
        d = np.abs(got - ref).max(-1)
        print(f"seed {seed} {' '.join(knobs) or 'default':24s}: max|dlogit| up to EOS {d[mask].max():.3e} mean {d[mask].mean():.2e}; ids equal {np.array_equal(ids.reshape(n,26)[mask], ref.argmax(-1)[mask])}; {dt*1e3:.1f} ms", flush=True)
