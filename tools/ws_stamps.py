"""GPU box tool: phase stamps (shader clock) of workgroup 0 of gemm_ws over its first 24 activation panels, waves 0 (early) and 4 (late)."""
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=False)
eng = Engine(d, precision="bf16")
assert eng.set_tuning(b"dec_stamps", 2) == 0
us = C.c_float()
M = int(sys.argv[1]) * 128 if len(sys.argv) > 1 else 1280 * 128
for flags in [int(x) for x in os.environ.get('WS_FLAGS', '0,1,2,3').split(',')]:
  eng.set_tuning(b"ws_dbg_flags", flags)
  print("== dbg flags", flags, "(1 = no stores, 2 = no loads)")
  for name, cout, act in (("qkv", 1152, 0), ("fc1", 1536, 2)):
      rc = eng.lib.ttr_bench_conv(eng.h, 1, 1, M, 384, 0, 1, 1, cout, act, 0, 3, C.byref(us))
      buf = (C.c_ulonglong * (26 * 16))()
      assert rc == 0 and eng.lib.ttr_dbg_dec_stamps(buf) == 0
      t = np.array(buf[:384], dtype=np.uint64).reshape(2, 24, 8).astype(np.float64)
      print(f"{name}: {us.value:.1f} us/launch; per-panel period (stamp0 deltas) wave0 {np.diff(t[0, 4:20, 0]).mean():.0f}  wave4 {np.diff(t[1, 4:20, 0]).mean():.0f} ticks")
      for w, lab in ((0, ["wait+barrier", "issue_x", "mfma", "epilogue"]), (1, ["wait+barrier", "issue_x", "epilogue(t-1)", "mfma"])):
          dt = np.diff(t[w, 4:20, :5], axis=1).mean(0)
          print(f"   wave {4 * w}: " + "  ".join(f"{l} {v:.0f}" for l, v in zip(lab, dt)))
