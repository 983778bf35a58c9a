"""GPU box tool: where and when the workgroups of qkv_attn4.hip's launch ran (100 MHz start / end, CU by HW_ID + XCC_ID): co-residency and balance.
   python3 tools/qkv_attn4_wgs.py [crops=1280]"""
import ctypes as C, os, sys, tempfile, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d)
assert eng.set_tuning(b"dec_stamps", 5) == 0
assert eng.set_tuning(b"qkv_attn4", 1) == 0
for k in sys.argv[2:]:
    a, b = k.split("="); assert eng.set_tuning(a.encode(), int(b)) == 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
crops = np.random.default_rng(0).integers(0, 256, (N, 32, 128, 3), dtype=np.uint8)
for _ in range(2):
    eng.parseq_logits(crops)
buf = (C.c_ulonglong * 4096)()
assert eng.lib.ttr_dbg_dec_stamps_ext(buf, 4096) == 0
w = np.array(buf[512:512 + 2048], dtype=np.uint64).reshape(512, 4).astype(np.int64)
w = w[w[:, 1] > 0]
t0 = w[:, 0].min()
start, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
hw, xcc = w[:, 2] & 0xFFFFFFFF, (w[:, 2] >> 32) & 0xF
mhz = w[:, 3] / (end - start)
cu = (xcc << 16) | (hw & 0xFF00)        # HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
print(f"{len(w)} workgroups of the last layer's launch; start {start.min():.1f} .. {start.max():.1f} us, end {end.min():.1f} .. {end.max():.1f} us, duration {np.mean(end - start):.1f} us (min {np.min(end - start):.1f}, max {np.max(end - start):.1f})")
by = collections.Counter(cu.tolist())
print(f"distinct CUs {len(by)}; workgroups per CU: {sorted(collections.Counter(by.values()).items())}")
late = start > 5.0
print(f"workgroups that started later than 5 us: {late.sum()}")
o = np.argsort(end)
print("slowest 8:", [(int(i), round(float(start[i]), 1), round(float(end[i]), 1), hex(int(cu[i]))) for i in o[-8:]])
print("fastest 8:", [(int(i), round(float(start[i]), 1), round(float(end[i]), 1), hex(int(cu[i]))) for i in o[:8]])
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  XCC {x}: {m.sum()} workgroups, blockIdx % 8 = {sorted(set((np.nonzero(w[:, 1] > 0)[0][m] % 8).tolist()))}, end {end[m].mean():.1f} us (min {end[m].min():.1f}, max {end[m].max():.1f}), shader clock {mhz[m].mean():.0f} MHz")
print("histogram of end times (us):", np.histogram(end, bins=8)[0].tolist(), np.round(np.histogram(end, bins=8)[1], 0).tolist())
