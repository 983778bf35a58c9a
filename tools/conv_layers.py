"""GPU box tool: CRAFT's 3x3 convolution layers alone (ttr_bench_conv, random data, 16 pages), each under a list of tuning
settings in ONE process, interleaved rounds (a device-to-device comparison is worth nothing: MI355X_MICROARCH.md, DVFS give-back 5).
  python tools/conv_layers.py [knob=a,b ...]      e.g.  python tools/conv_layers.py c3_order=0,1"""
import ctypes as C, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=False)
eng = Engine(d, precision="bf16")
knob, vals = "c3_order", [0]
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    knob, vals = k, [int(x) for x in v.split(",")]
us = C.c_float()
P = int(os.environ.get("PAGES", "16"))
LAYERS = [  # name, B, H, W, C0, C1, ks, dil, Cout, act, f32resid
    ("slice1.7", P, 512, 384, 64, 0, 3, 1, 128, 1, 0), ("slice1.10", P, 512, 384, 128, 0, 3, 1, 128, 1, 0),
    ("slice2.14", P, 256, 192, 128, 0, 3, 1, 256, 1, 0), ("slice2.17", P, 256, 192, 256, 0, 3, 1, 256, 1, 0),
    ("slice3.24", P, 128, 96, 256, 0, 3, 1, 512, 1, 0), ("slice3.27", P, 128, 96, 512, 0, 3, 1, 512, 1, 0),
    ("slice4.34", P, 64, 48, 512, 0, 3, 1, 512, 1, 0), ("upconv1.3", P, 64, 48, 512, 0, 3, 1, 256, 1, 0),
    ("upconv2.3", P, 128, 96, 256, 0, 3, 1, 128, 1, 0), ("upconv3.3", P, 256, 192, 128, 0, 3, 1, 64, 1, 0),
    ("upconv4.3", P, 512, 384, 64, 0, 3, 1, 32, 1, 0),
]
rounds = int(os.environ.get("ROUNDS", "5"))
res = {(n[0], v): [] for n in LAYERS for v in vals}
for r in range(rounds):
    for L in LAYERS:
        for v in vals:
            assert eng.set_tuning(knob, v) == 0
            rc = eng.lib.ttr_bench_conv(eng.h, *L[1:], 3, C.byref(us))
            assert rc == 0, (L, rc)
            res[(L[0], v)].append(us.value)
tot = {v: 0.0 for v in vals}
for L in LAYERS:
    fl = 2.0 * L[1] * L[2] * L[3] * L[8] * 9 * L[4]
    row = []
    for v in vals:
        m = float(np.median(res[(L[0], v)])); tot[v] += m
        row.append(f"{knob}={v}: {m:7.1f} us {fl / m / 1e9:6.2f} PF/s")
    print(f"{L[0]:10s} " + "   ".join(row))
print("sum        " + "   ".join(f"{knob}={v}: {tot[v]:7.1f} us" for v in vals))
