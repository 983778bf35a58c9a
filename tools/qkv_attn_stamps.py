"""GPU box tool: phase stamps (shader clock, wave 0 of workgroup 0, tiles 4..20) of the fused qkv + attention launch at 1280 crops: where a tile's time goes.
   python3 tools/qkv_attn_stamps.py [crops=1280] [four=0]     (four = 1: qkv_attn4.hip's four-wave tiles, two workgroups per CU)"""
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d)
assert eng.set_tuning(b"dec_stamps", 5) == 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
FOUR = int(sys.argv[2]) if len(sys.argv) > 2 else 0
assert eng.set_tuning(b"qkv_attn4", FOUR) == 0
for kv in sys.argv[3:]:
    a, b = kv.split("="); assert eng.set_tuning(a.encode(), int(b)) == 0
crops = np.random.default_rng(0).integers(0, 256, (N, 32, 128, 3), dtype=np.uint8)
for _ in range(2):
    eng.parseq_logits(crops)
buf = (C.c_ulonglong * (26 * 16))()
assert eng.lib.ttr_dbg_dec_stamps(buf) == 0
t = np.array(buf[:24 * 16], dtype=np.uint64).reshape(24, 16).astype(np.float64)
lab = ["K loop", "convert Q K V + write Q K", "barrier 1", "S = Q K^T", "softmax + P planes", "barrier 2", "write V + barrier 3", "P V", "wait vmcnt(0)", "store + drain"]
if FOUR:
    lab = ["K loop", "barrier 0", "convert Q K V + write Q K", "barrier 1", "S = Q K^T", "softmax + P planes", "barrier 2 + write V + barrier 3", "P V", "barrier 4 + request", "store"]
sel = t[4:20] if not FOUR else t[3:13]        # (four-wave tiles: a workgroup has half as many)
d = np.diff(sel[:, :11], axis=1)
period = np.diff(sel[:, 0])
print(f"{N} crops: tile period {period.mean():.0f} ticks (min {period.min():.0f}, max {period.max():.0f}); phases of a tile (mean ticks, share):")
for l, v in zip(lab, d.mean(0)):
    print(f"  {l:28s} {v:8.0f}  {100 * v / d.mean(0).sum():5.1f} %")
real = np.diff(sel[:, 11])                        # 100 MHz ticks per tile
print(f"  shader clock over these tiles: {period.sum() / real.sum() * 100:.0f} MHz; tile period {real.mean() / 100:.1f} us")
print(f"  sum {d.mean(0).sum():.0f}; next tile's head (stamp 10 -> next stamp 0): {(sel[1:, 0] - sel[:-1, 10]).mean():.0f}")
