"""GPU box tool: where the host time of one batch goes (engine's own wall-clock splits + the Python result hand-over).

  python tools/host_gaps.py [--pages 32] [--words 28] [--steps 6]"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import synth, weights as W  # noqa: E402
from tuatara_amd.engine import DeviceBuffer, Engine  # noqa: E402
import ctypes as C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=32)
    ap.add_argument("--words", type=int, default=28)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--profiling", type=int, default=1)
    a = ap.parse_args()
    d = tempfile.mkdtemp()
    W.make_synthetic_weights(d, seed=0, structured=True)
    eng = Engine(d, precision="bf16")
    H, Wd, P = 1024, 768, a.pages
    pages = np.stack([synth.synthetic_page(i, H, Wd, n_words=a.words) for i in range(P)])
    buf = DeviceBuffer(pages.nbytes)
    buf.upload(pages)
    eng.set_profiling(a.profiling)
    names = ["enqueue", "wait_counters", "wait_cands", "calipers", "rects+parseq_enq", "wait_gpu", "events", "decode"]
    for s in range(a.steps):
        t0 = time.perf_counter()
        arr = (C.c_void_p * P)()
        eng._check(eng.lib.ttr_pages_to_data_dev(eng.h, buf.ptr, P, H, Wd, arr))
        t1 = time.perf_counter()
        res = eng._take_many(arr, P)
        t2 = time.perf_counter()
        us = eng.last_host_us()
        print(f"step {s}: C call {1e3 * (t1 - t0):.3f} ms, python take {1e3 * (t2 - t1):.3f} ms, stages {eng.last_stage_ms()}")
        print("   host us:", dict(zip(names, us)), "sum", round(sum(us), 1))


if __name__ == "__main__":
    main()
