"""GPU box tool: phase stamps of mlp_fused (workgroup 0, wave 0, first panel) + launch time."""
import ctypes as C, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=False)
eng = Engine(d, precision="bf16")
assert eng.set_tuning(b"dec_stamps", 3) == 0
M = (int(sys.argv[1]) if len(sys.argv) > 1 else 1280) * 128
rng = np.random.default_rng(0)
x = rng.standard_normal((M, 384)).astype(np.float32)
g = np.ones(384, np.float32); b = np.zeros(384, np.float32)
w1 = (rng.standard_normal((1536, 384)) / 20).astype(np.float32); b1 = np.zeros(1536, np.float32)
w2 = (rng.standard_normal((384, 1536)) / 40).astype(np.float32); b2 = np.zeros(384, np.float32)
proj = len(sys.argv) > 2 and sys.argv[2] == "proj"
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    assert eng.set_tuning(k, int(v)) == 0, kv
if proj:
    att = rng.standard_normal((M, 384)).astype(np.float32); wp = (rng.standard_normal((384, 384)) / 20).astype(np.float32)
    eng.dbg_mlp(x, g, b, w1, b1, w2, b2, g, b, att=att, wp=wp, bp=b)
else:
    eng.dbg_mlp(x, g, b, w1, b1, w2, b2, g, b)
buf = (C.c_ulonglong * (26 * 16))()
assert eng.lib.ttr_dbg_dec_stamps(buf) == 0
t = np.array(buf[:384], dtype=np.uint64).reshape(48, 8).astype(np.float64)
names = ["wait+barrier", "issue 12 DMA", "GEMM1", "GELU", "GEMM2"]
dt = np.diff(t[4:44, :6], axis=1)
print("per-chunk period", np.diff(t[4:44, 0]).mean(), "cycles")
for n, v in zip(names, dt.mean(0)):
    print(f"   {n:14s} {v:8.0f}")
tp = np.array(buf[384:400], dtype=np.uint64).reshape(4, 4).astype(np.float64)
tw = np.array(buf[400:416], dtype=np.uint64).reshape(4, 4).astype(np.float64)
for k in range(4):
    a = tp[k]
    if a[3] > a[0] > 0:
        print(f"panel {k}: front (LayerNorm / projection) {a[1]-a[0]:.0f}  chunk loop {a[2]-a[1]:.0f}  epilogue {a[3]-a[2]:.0f}  cycles; {(tw[k][3]-tw[k][0])/100:.1f} us, clock {(a[3]-a[0])/max(tw[k][3]-tw[k][0],1)*0.1:.2f} GHz" + (f"  gap to next {tp[k+1][0]-a[3]:.0f}" if k < 3 and tp[k+1][0] > 0 else ""))
