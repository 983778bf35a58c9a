"""GPU box tool: the same 32-page batch through the whole path N times (bf16 engine, streamed and synchronous), every result compared
with the first: the path has no run-to-run freedom (integer atomics in the CCL are order-free), so any difference is a race."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import synth, weights as W
from tuatara_amd.engine import DeviceBuffer, Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d, precision=os.environ.get("TTR_PREC", "bf16"))
P = 32
pages = np.stack([synth.synthetic_page(200 + i, 1024, 768, n_words=40) for i in range(P)])
buf = DeviceBuffer(pages.nbytes); buf.upload(pages)
def key(res): return [[(tuple(x["bbox"]), x["text"]) for x in pg] for pg in res]
ref = key(eng.pages_to_data_dev(buf, P, 1024, 768))
print("crops", sum(len(p) for p in ref))
bad = 0
for it in range(n):
    got = key(eng.pages_to_data_dev(buf, P, 1024, 768))
    if got != ref:
        bad += 1
        print("iteration", it, "differs on pages", [i for i in range(P) if got[i] != ref[i]][:8])
prev = None
for it in range(n):                      # streamed: results of push k come back at push k + 1
    out = eng.stream_push(buf, P, 1024, 768)
    if out and key(out) != ref:
        bad += 1; print("streamed iteration", it, "differs")
out = eng.stream_flush()
if out and key(out) != ref:
    bad += 1; print("streamed flush differs")
print("soak:", n, "synchronous +", n, "streamed passes,", bad, "differences")
sys.exit(1 if bad else 0)
