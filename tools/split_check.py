"""GPU box: the f16x4 (split-operand) engine against the CPU fp32 oracle and the fp32-MFMA engine.
   python tools/split_check.py [n_crops]   (development tool; uses the oracle)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tuatara_amd import weights as W
from tuatara_amd.build import build_lib
from tuatara_amd.engine import Engine
from oracle import pipeline
from tests import parity_rules as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
build_lib()
wdir = "/tmp/split_check_w"
c, p = W.make_synthetic_weights(wdir, seed=0, structured=True)
craft, parseq = pipeline.load_models(c, p)
crops = np.random.default_rng(0).integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
with torch.no_grad():
    x = torch.from_numpy(crops).permute(0, 3, 1, 2).float().div(255.0)
    ref, ref_ar = parseq(x, return_ar=True)
ref, ref_ar = ref.numpy(), ref_ar.numpy()
up = R.upto_eos(ref.argmax(-1))
mask = np.arange(26)[None, :] < up[:, None]
for prec in ("f32", "f16x4"):
    e = Engine(wdir, precision=prec)
    got, got_ar, ids = e.parseq_logits(crops, want_ar=True)
    t = time.time()
    for _ in range(3): e.parseq_logits(crops)
    dt = (time.time() - t) / 3
    d = np.abs(got - ref).max(-1); dar = np.abs(got_ar - ref_ar).max(-1)
    print(f"{prec:6s} PARSeq {n} crops: refined max|dlogit| up to EOS {d[mask].max():.3e} (all {d.max():.3e}, mean {d[mask].mean():.2e}); AR {dar[mask].max():.3e}; "
          f"ids equal {np.array_equal(ids.reshape(n, 26)[mask], ref.argmax(-1)[mask])}; {dt*1e3:.1f} ms per forward", flush=True)
    del e
# CRAFT: FUNSD canvas (structured weights) + a random canvas with fully random weights, then FUNSD end to end
from oracle import post
from PIL import Image
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
funsd = np.array(Image.open(os.path.join(root, "tests", "data", "funsd_0001129658.png")).convert("RGB"))
canvas, _ = post.resize_aspect_ratio(np.ascontiguousarray(funsd[:, :, ::-1]))
ref_heat = pipeline.craft_heatmap(craft, canvas)
ref_e2e = pipeline.image_to_data(craft, parseq, funsd)
wdir_r = "/tmp/split_check_wr"
cr, pr = W.make_synthetic_weights(wdir_r, seed=0, structured=False)
craft_r, _ = pipeline.load_models(cr, pr)
canvas_r = np.random.default_rng(7).integers(0, 256, (256, 192, 3), dtype=np.uint8)
ref_heat_r = pipeline.craft_heatmap(craft_r, canvas_r)
for prec in ("f32", "f16x4"):
    e = Engine(wdir, precision=prec)
    got = e.craft_heatmap(canvas)
    t = time.time()
    for _ in range(3): e.craft_heatmap(canvas)
    dt = (time.time() - t) / 3
    out = e.image_to_data(funsd)
    same_box = [o["bbox"] for o in out] == [list(o["bbox"]) for o in ref_e2e] if len(out) == len(ref_e2e) else False
    same_txt = [o["text"] for o in out] == [o["text"] for o in ref_e2e]
    print(f"{prec:6s} CRAFT FUNSD canvas: max|dheat| {np.abs(got - ref_heat).max():.3e}; {dt*1e3:.1f} ms per page; e2e {len(out)} boxes (oracle {len(ref_e2e)}), bboxes equal {same_box}, strings equal {same_txt}", flush=True)
    del e
    e = Engine(wdir_r, precision=prec)
    got = e.craft_heatmap(canvas_r)
    print(f"{prec:6s} CRAFT random weights 256x192: max|dheat| {np.abs(got - ref_heat_r).max():.3e} (max|ref| {np.abs(ref_heat_r).max():.2f})", flush=True)
    del e
