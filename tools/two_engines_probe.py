"""GPU box probe: does a second engine (own stream, own workspaces) on the SAME GPU raise the page rate?  Each engine is fed streamed
batches from its own host thread, so the GPU sees two independent queues: one queue's HBM-bound kernels and tile tails can run under the
other's matrix-bound convolutions.   python tools/two_engines_probe.py [engines] [passes per engine] [pages per batch] [key=value ...]"""
import os, sys, tempfile, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import synth, weights as W
from tuatara_amd.engine import DeviceBuffer, Engine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
P = int(sys.argv[3]) if len(sys.argv) > 3 else 32
d = tempfile.mkdtemp()
W.make_synthetic_weights(d, seed=0, structured=True)
NB = 4
bufs = []
for b in range(NB):
    pages = np.stack([synth.synthetic_page((b * P + i) % 512, 1024, 768, n_words=40, layout="cells5x8") for i in range(P)])
    buf = DeviceBuffer(pages.nbytes)
    buf.upload(pages)
    bufs.append(buf)
engs = [Engine(d, precision="f16x4", bench_grid_boxes=1) for _ in range(E)]
for e in engs:
    for kv in sys.argv[4:]:
        k, v = kv.split("=")
        assert e.set_tuning(k, int(v)) == 0, kv


def feed(e, k, off):
    for j in range(k):
        e.stream_push(bufs[(off + j) % NB], P, 1024, 768, keep=False)
    while e.stream_flush(keep=False):
        pass


def run(k):
    ts = [threading.Thread(target=feed, args=(e, k, i)) for i, e in enumerate(engs)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in engs:
        e.lib.ttr_dev_sync(e.h)
    return time.perf_counter() - t0


run(3)
dt = run(passes)
print(f"engines {E} pages/batch {P} passes/engine {passes}: {E * passes * P / dt:.1f} pages/s ({dt / (E * passes) * 1e3:.2f} ms per batch)")
