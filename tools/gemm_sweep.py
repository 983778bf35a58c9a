"""GPU box tool: check every gemm2 tile configuration against a float64 numpy convolution and time
the real CRAFT / PARSeq layer shapes under each configuration (and under the first-generation igemm).

  python tools/gemm_sweep.py [--pages 8] [--crops 320] [--iters 20] [--skip-check]
Writes a table to stdout (redirect into gpurun_out/)."""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import weights as W  # noqa: E402
from tuatara_amd.engine import Engine  # noqa: E402


def bf16_round(a: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def conv_ref(x, w, bias, ks, dil, act):
    B, H, Wd, Cc = x.shape
    Cout = w.shape[0]
    x = bf16_round(x).astype(np.float64)
    w = bf16_round(w).astype(np.float64)
    pad = dil * (ks // 2)
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    out = np.zeros((B, H, Wd, Cout))
    for ky in range(ks):
        for kx in range(ks):
            out += np.einsum("bhwc,oc->bhwo", xp[:, ky * dil:ky * dil + H, kx * dil:kx * dil + Wd], w[:, ky, kx])
    if bias is not None:
        out += bias.astype(np.float64)
    if act == 1:
        out = np.maximum(out, 0)
    elif act == 2:
        from scipy.special import erf
        out = 0.5 * out * (1 + erf(out / np.sqrt(2)))
    return out


CHECKS = [  # B, H, W, C0, C1, ks, dil, Cout, act
    (1, 20, 24, 64, 0, 3, 1, 64, 1),
    (2, 13, 17, 128, 0, 3, 1, 136, 0),      # ragged M, ragged Cout, batch boundary halos
    (1, 16, 20, 64, 0, 3, 6, 256, 0),       # dilation 6
    (1, 9, 31, 128, 64, 1, 1, 128, 1),      # virtual concat, 1x1
    (1, 12, 12, 64, 64, 3, 1, 72, 2),       # concat + 3x3 + gelu + ragged Cout
    (1, 1, 700, 384, 0, 1, 1, 1152, 0),     # linear layer
    (1, 1, 300, 1536, 0, 1, 1, 384, 2),
    (1, 16, 64, 64, 0, 3, 1, 64, 1),        # conv3p-eligible shapes (H % 8, W % 32)
    (2, 8, 32, 128, 0, 3, 1, 136, 0),
    (1, 24, 96, 64, 0, 3, 1, 256, 1),
]

LAYERS = [  # name, per-page H, W (0 = PARSeq rows), C0, C1, ks, dil, Cout, act, f32_resid
    ("slice1.3", 1024, 768, 64, 0, 3, 1, 64, 1, 0),
    ("slice1.7", 512, 384, 64, 0, 3, 1, 128, 1, 0),
    ("slice1.10", 512, 384, 128, 0, 3, 1, 128, 0, 0),
    ("slice2.14", 256, 192, 128, 0, 3, 1, 256, 1, 0),
    ("slice2.17", 256, 192, 256, 0, 3, 1, 256, 0, 0),
    ("slice3.24", 128, 96, 256, 0, 3, 1, 512, 1, 0),
    ("slice3.27", 128, 96, 512, 0, 3, 1, 512, 0, 0),
    ("slice4.34", 64, 48, 512, 0, 3, 1, 512, 1, 0),
    ("slice5.1", 64, 48, 512, 0, 3, 6, 1024, 0, 0),
    ("slice5.2", 64, 48, 1024, 0, 1, 1, 1024, 0, 0),
    ("upconv1.0", 64, 48, 1024, 512, 1, 1, 512, 1, 0),
    ("upconv1.3", 64, 48, 512, 0, 3, 1, 256, 1, 0),
    ("upconv2.0", 128, 96, 256, 512, 1, 1, 256, 1, 0),
    ("upconv2.3", 128, 96, 256, 0, 3, 1, 128, 1, 0),
    ("upconv3.0", 256, 192, 128, 256, 1, 1, 128, 1, 0),
    ("upconv3.3", 256, 192, 128, 0, 3, 1, 64, 1, 0),
    ("upconv4.0", 512, 384, 64, 128, 1, 1, 64, 1, 0),
    ("upconv4.3", 512, 384, 64, 0, 3, 1, 32, 1, 0),
    ("pq.qkv", 0, 0, 384, 0, 1, 1, 1152, 0, 0),
    ("pq.proj", 0, 0, 384, 0, 1, 1, 384, 0, 1),
    ("pq.fc1", 0, 0, 384, 0, 1, 1, 1536, 2, 0),
    ("pq.fc2", 0, 0, 1536, 0, 1, 1, 384, 0, 1),
    ("pq.crosskv", 0, 0, 384, 0, 1, 1, 768, 0, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=8)
    ap.add_argument("--crops", type=int, default=320)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--cfgs", default="-1,1,2,3,4,5,6,7")
    ap.add_argument("--tune", action="append", default=[], help="ttr_set_tuning key=value, repeatable")
    args = ap.parse_args()
    cfgs = [int(c) for c in args.cfgs.split(",")]

    d = tempfile.mkdtemp()
    W.make_synthetic_weights(d, seed=0, structured=False)
    eng = Engine(d, precision="bf16")
    rng = np.random.default_rng(0)
    for kv in args.tune:
        k, v = kv.split("=")
        assert eng.set_tuning(k.encode(), int(v)) == 0, kv

    if not args.skip_check:
        print("== correctness (max |err| / max |ref|) ==")
        worst = 0.0
        for (B, H, Wd, C0, C1, ks, dil, Cout, act) in CHECKS:
            x0 = rng.standard_normal((B, H, Wd, C0)).astype(np.float32)
            x1 = rng.standard_normal((B, H, Wd, C1)).astype(np.float32) if C1 else None
            w = (rng.standard_normal((Cout, ks, ks, C0 + C1)) / np.sqrt(ks * ks * (C0 + C1))).astype(np.float32)
            bias = rng.standard_normal(Cout).astype(np.float32)
            ref = conv_ref(np.concatenate([x0, x1], -1) if C1 else x0, w, bias, ks, dil, act)
            row = []
            for cfg in cfgs:
                eng.lib.ttr_set_gemm_config(cfg)
                got = eng.dbg_conv(x0, w, bias, ks=ks, dil=dil, act=act, x1=x1)
                err = float(np.abs(got - ref).max() / np.abs(ref).max())
                worst = max(worst, err)
                row.append(f"{cfg}:{err:.1e}")
            print(f"B{B} {H}x{Wd} C{C0}+{C1} k{ks} d{dil} -> {Cout} act{act}: " + " ".join(row), flush=True)
        print("worst", worst, "OK" if worst < 2e-5 else "FAIL")

    print(f"== timing: us per launch, TFLOP/s (pages={args.pages}, crops={args.crops}) ==")
    us = C.c_float()
    for (name, H, Wd, C0, C1, ks, dil, Cout, act, f32r) in LAYERS:
        if H:
            B, hh, ww = args.pages, H, Wd
        else:
            B, hh, ww = 1, 1, args.crops * 128
        flops = 2.0 * B * hh * ww * Cout * ks * ks * (C0 + C1)
        row = []
        for cfg in cfgs:
            eng.lib.ttr_set_gemm_config(cfg)
            rc = eng.lib.ttr_bench_conv(eng.h, B, hh, ww, C0, C1, ks, dil, Cout, act, f32r, args.iters, C.byref(us))
            if rc != 0:
                row.append(f"{cfg}: ERR {eng.lib.ttr_last_error().decode()}")
                continue
            row.append(f"{cfg}:{us.value:8.1f}us {flops / us.value / 1e6:6.0f}T")
        print(f"{name:10s} " + " | ".join(row), flush=True)
    eng.lib.ttr_set_gemm_config(0)


if __name__ == "__main__":
    main()
