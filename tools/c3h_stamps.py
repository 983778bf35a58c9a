"""GPU box tool: phase stamps (shader clock) of workgroup 0, wave 0 of conv3h_kernel (CRAFT's packed-pairs head, persistent) over its first 24 patches
(the last head launch of the pass: conv_cls.4 + tail).   python tools/c3h_stamps.py [pages] [key=value ...]"""
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import synth, weights as W
from tuatara_amd.engine import DeviceBuffer, Engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d, precision="f16x4")
assert eng.set_tuning(b"dec_stamps", 6) == 0
assert eng.set_tuning(b"detector_only", 1) == 0
for kv in sys.argv[2:]:
    k, v = kv.split("="); assert eng.set_tuning(k, int(v)) == 0, kv
pages = np.stack([synth.synthetic_page(i, 1024, 768, n_words=28) for i in range(P)])
buf = DeviceBuffer(pages.nbytes); buf.upload(pages)
for s in range(2): eng.pages_to_data_dev(buf, P, 1024, 768)
out = (C.c_ulonglong * (26 * 16))()
assert eng.lib.ttr_dbg_dec_stamps(out) == 0
t = np.array(out[:192], dtype=np.uint64).reshape(24, 8).astype(np.float64)
n = int((t[:, 0] > 0).sum())
print(f"{n} patches stamped; per-patch period {np.diff(t[2:n-1, 0]).mean():.0f} ticks (100 MHz: x 10 ns)")
lab = ["wait for the patch", "barrier", "216 MFMAs (+ fragment reads)", "barrier", "request next patch", "epilogue issue", "-> next patch top"]
dt = np.diff(t[2:n-1, :7], axis=1).mean(0)
nxt = (t[3:n, 0] - t[2:n-1, 6]).mean()
for l, v in zip(lab, list(dt) + [nxt]): print(f"   {l:32s} {v:8.0f}")
