"""GPU box tool: phase stamps (shader clock) of workgroup 0 of conv3p_first2 (CRAFT conv1_1 + conv1_2) over its first 24 patches,
waves 0 (MFMA first) and 4 (prologue first).   python tools/first2_stamps.py [pages]"""
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W
from tuatara_amd.engine import DeviceBuffer, Engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d, precision="bf16")
assert eng.set_tuning(b"dec_stamps", 4) == 0
assert eng.set_tuning(b"c3_first_persistent", int(os.environ.get("F2_MODE", "2"))) == 0
pages = np.full((P, 1024, 768, 3), 255, np.uint8)
buf = DeviceBuffer(pages.nbytes); buf.upload(pages)
for s in range(2): eng.pages_to_data_dev(buf, P, 1024, 768)
out = (C.c_ulonglong * (26 * 16))()
assert eng.lib.ttr_dbg_dec_stamps(out) == 0
t = np.array(out[:384], dtype=np.uint64).reshape(2, 24, 8).astype(np.float64)
mode = int(os.environ.get("F2_MODE", "2"))
if mode == 2:   # wave-specialised kernel: wave 0 = consumer, wave 4 = producer
    for w, lab in ((0, ["barrier", "288 MFMAs", "pool epilogue"]), (1, ["barrier", "canvas strip", "tile 0 gather", "tile 0 rest", "tiles 1-2", "tiles 3-5"])):
        print(f"wave {4 * w}: per-patch period {np.diff(t[w, 4:20, 0]).mean():.0f} ticks")
        dt = np.diff(t[w, 4:20, :len(lab) + 1], axis=1).mean(0)
        print("   " + "  ".join(f"{l} {v:.0f}" for l, v in zip(lab, dt)))
else:
    for w in (0, 1):
        print(f"wave {4 * w}: per-patch period {np.diff(t[w, 4:20, 0]).mean():.0f} ticks")
        lab = ["barrier A", "cv write + canvas issue", "barrier B", "mfma" if w == 0 else "prologue", "prologue" if w == 0 else "mfma", "epilogue"]
        dt = np.diff(t[w, 4:20, :7], axis=1).mean(0)
        print("   " + "  ".join(f"{l} {v:.0f}" for l, v in zip(lab, dt)))
