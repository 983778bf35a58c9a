"""GPU box tool for rocprofv3 --pmc runs: a few representative layers, a handful of launches each (random data).
  rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d out -- python3 tools/pmc_layers.py"""
import ctypes as C, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=False)
eng = Engine(d, precision="bf16")
us = C.c_float()
M = int(os.environ.get("PMC_CROPS", "614")) * 128
ONLY = os.environ.get("PMC_ONLY", "")
LAYERS = [  # name, B, H, W, C0, C1, ks, dil, Cout, act, f32resid
    ("vit.qkv", 1, 1, M, 384, 0, 1, 1, 1152, 0, 0), ("vit.proj", 1, 1, M, 384, 0, 1, 1, 384, 0, 1),
    ("vit.fc1", 1, 1, M, 384, 0, 1, 1, 1536, 2, 0), ("vit.fc2", 1, 1, M, 1536, 0, 1, 1, 384, 0, 1),
    ("gemm.K1536.N1536", 1, 1, M, 1536, 0, 1, 1, 1536, 0, 0),
    ("craft.slice3.27", 16, 128, 96, 512, 0, 3, 1, 512, 0, 0), ("craft.slice2.17", 16, 256, 192, 256, 0, 3, 1, 256, 0, 0),
    ("craft.slice1.10", 16, 512, 384, 128, 0, 3, 1, 128, 0, 0), ("craft.slice1.3", 16, 1024, 768, 64, 0, 3, 1, 64, 1, 0),
    ("craft.slice5.1", 16, 64, 48, 512, 0, 3, 6, 1024, 0, 0),
]
for (name, B, H, Wd, C0, C1, ks, dil, Cout, act, f32r) in LAYERS:
    if ONLY and not name.startswith(ONLY):
        continue
    rc = eng.lib.ttr_bench_conv(eng.h, B, H, Wd, C0, C1, ks, dil, Cout, act, f32r, 3, C.byref(us))
    print(name, rc, round(us.value, 1), "us", round(2.0 * B * H * Wd * Cout * ks * ks * (C0 + C1) / us.value / 1e6), "TF", flush=True)
