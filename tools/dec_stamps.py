"""GPU box tool: where does a step of the fused AR decoder go?  Shader-clock stamps of workgroup 0 per phase."""
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W
from tuatara_amd.engine import Engine
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d, precision="bf16")
names = ["embed+LNc", "self_kv gemm", "self attn", "self_out gemm + LN1", "cross_q gemm", "cross attn", "cross_out gemm + LN2", "ffn1", "ffn2 + LNf", "head", "argmax"]
crops = np.random.default_rng(0).integers(0, 256, (int(sys.argv[1]) if len(sys.argv) > 1 else 614, 32, 128, 3), dtype=np.uint8)
eng.set_tuning(b"dec_stamps", 1)
for G in (4, 8, 16):
    eng.set_tuning(b"decoder_mode", G)
    eng.parseq_logits(crops); eng.parseq_logits(crops)
    buf = (C.c_ulonglong * (26 * 16))()
    assert eng.lib.ttr_dbg_dec_stamps(buf) == 0
    t = np.array(buf[:], dtype=np.uint64).reshape(26, 16).astype(np.float64)
    dt = np.diff(np.concatenate([t[:25, :11], t[1:26, :1]], 1), axis=1)[3:24]    # steady-state steps
    print(f"G={G}: cycles per phase (mean over steps 3..23), total {dt.sum(1).mean():.0f} cycles/step")
    for n, v in zip(names, dt.mean(0)):
        print(f"   {n:24s} {v:9.0f}")
