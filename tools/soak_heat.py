"""GPU box tool: run-to-run determinism of the detector's heat map, bit for bit (the default engine; fully random CRAFT weights so that every layer matters):
the same canvases N times, every heat map compared with the first.  The kernels with counted waits (conv1_split, conv3h.hip, conv1u.hip) have no run-to-run freedom;
any difference is a race.   python3 tools/soak_heat.py [iterations]"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
d = tempfile.mkdtemp()
W.export_craft(W.synth_craft(3, False), d); W.export_parseq(W.synth_parseq(0), d)
eng = Engine(d)
rng = np.random.default_rng(9)
bad = 0
for hw in ((1024, 768), (512, 1024), (256, 192), (96, 160)):
    cv = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    ref = eng.craft_heatmap(cv)
    for it in range(n):
        got = eng.craft_heatmap(cv)
        if not np.array_equal(got, ref):
            bad += 1
            print(hw, "iteration", it, "differs:", int((got != ref).sum()), "values, max", float(np.abs(got - ref).max()))
print("soak_heat:", n, "iterations x 4 canvases,", bad, "differences")
sys.exit(1 if bad else 0)
