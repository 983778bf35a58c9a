"""GPU box tool: run CRAFT alone (bf16) on synthetic 1024x768 pages, one page per call, so a rocprofv3 --pmc pass sees
only the detector's kernels.  python tools/prof_craft.py [pages]"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import synth, weights as W
from tuatara_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d = tempfile.mkdtemp()
W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d, precision="bf16")
for i in range(n):
    pg = synth.synthetic_page(i, 1024, 768, n_words=28)
    heat = eng.craft_heatmap(pg)
print("craft pages:", n, heat.shape, float(heat.mean()))
