"""GPU box tool for rocprofv3 --pmc passes: the whole hot path on one batch of synthetic 1024x768 pages (default 16 = one CRAFT
group), a few steps, nothing else.   python tools/prof_pages.py [pages] [steps] [words per page]"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import synth, weights as W
from tuatara_amd.engine import DeviceBuffer, Engine

P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
words = int(sys.argv[3]) if len(sys.argv) > 3 else 28   # 0: blank pages, no crops -> the detector only
d = tempfile.mkdtemp()
W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d, precision=os.environ.get("TTR_PREC", "bf16"))
for kv in sys.argv[4:]:                      # key=value tuning knobs (Engine.set_tuning)
    k, v = kv.split("=")
    assert eng.set_tuning(k, int(v)) == 0, kv
if not words:
    assert eng.set_tuning("detector_only", 1) == 0
pages = np.stack([synth.synthetic_page(i, 1024, 768, n_words=words) for i in range(P)]) if words else np.full((P, 1024, 768, 3), 255, np.uint8)
buf = DeviceBuffer(pages.nbytes)
buf.upload(pages)
for s in range(steps):
    res = eng.pages_to_data_dev(buf, P, 1024, 768)
print("pages", P, "steps", steps, "crops/page", float(np.mean([len(r) for r in res])))
