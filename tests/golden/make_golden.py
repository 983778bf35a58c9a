"""Generate the committed golden fixtures from the CPU oracle (run in the build container).

The reference has no tests, golden vectors or runnable build here, so these
vectors pin *the oracle's* behaviour (and, for the tokenizer table, the behaviour derived
by reading tuatara.cpp:31-48): they guard the restatement against regressions and travel to
the GPU box, where the engine is checked against them without re-running the slow CPU models.

  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import pipeline, post  # noqa: E402
from tuatara_amd import weights as W  # noqa: E402


def synthetic_heatmap(seed: int, H: int = 256, W_: int = 192):
    """G4: text lines of anisotropic Gaussian 'word' blobs with affinity bridges between some
    neighbours; includes edge-touching words, rotated words, weak words (peak < text_threshold),
    link-only bridges and specks with area < 10."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W_].astype(np.float32)
    text = np.zeros((H, W_), np.float32)
    link = np.zeros((H, W_), np.float32)
    k = 0
    for line_y in np.arange(10, H - 4, 22.0):
        x = rng.uniform(-6, 10)
        prev = None
        while x < W_ + 4:
            sx, sy = rng.uniform(5, 13), rng.uniform(1.8, 3.2)
            cx, cy = x + 2 * sx, line_y + rng.uniform(-2, 2)
            th = rng.uniform(-0.35, 0.35) if k % 4 == 0 else 0.0
            amp = 0.5 if k % 7 == 3 else rng.uniform(0.85, 1.3)
            dx, dy = xx - cx, yy - cy
            u, v = dx * np.cos(th) + dy * np.sin(th), -dx * np.sin(th) + dy * np.cos(th)
            text += amp * np.exp(-0.5 * ((u / sx) ** 2 + (v / sy) ** 2))
            if prev is not None and k % 3 != 0:  # affinity bridge to the previous word
                mx, my = (prev[0] + cx) / 2, (prev[1] + cy) / 2
                link += 0.9 * np.exp(-0.5 * (((xx - mx) / (abs(cx - prev[0]) / 3 + 1)) ** 2 + ((yy - my) / 1.8) ** 2))
            prev = (cx, cy)
            x = cx + 2 * sx + rng.uniform(6, 14)
            k += 1
    for _ in range(5):  # specks (area < 10)
        x, y = int(rng.uniform(2, W_ - 3)), int(rng.uniform(2, H - 2))
        text[y, x:x + 2] += 0.95
    text = np.minimum(text, 1.0)
    link = np.minimum(link, 1.0)
    text += rng.normal(0, 0.01, text.shape).astype(np.float32)
    link += rng.normal(0, 0.01, link.shape).astype(np.float32)
    return np.stack([text, link], -1).astype(np.float32)


def main():
    out = {}
    # G1 tokenizer
    itos, eos, bos, pad = post.tokenizer_table()
    out["tokenizer"] = {"itos": itos, "eos_id": eos, "bos_id": bos, "pad_id": pad,
                        "cases": [[ids, post.decode_ids(ids)] for ids in [
                            [1, 2, 3, 0, 4], [88, 11, 88, 12, 0], [69, 70, 71, 79, 76, 78, 75, 77, 0], [0], [94, 93, 92, 91, 90, 89, 87, 86],
                            [37, 38, 88, 88, 0, 5], list(range(60, 95))]]}
    # G2 resize_aspect_ratio dims
    out["resize_dims"] = [[h, w, list(post.resize_aspect_ratio_dims(h, w))] for h, w in
                          [(1000, 754), (1000, 814), (763, 607), (206, 275), (664, 1245), (768, 1024), (1024, 768), (2000, 1500), (2048, 1536), (31, 17)]]
    # G3 niter table is covered by G4 boxes; G4 synthetic heat maps -> rects
    g4 = {}
    for seed in (0, 1, 2):
        heat = synthetic_heatmap(seed)
        rects, labels, _ = post.get_detected_boxes(heat[..., 0], heat[..., 1])
        g4[f"rects_{seed}"] = rects
        g4[f"nlabels_{seed}"] = np.array([labels.max()])
    np.savez_compressed(os.path.join(HERE, "g4_boxes.npz"), **g4)
    # G5 PARSeq logits, G6 CRAFT small heat map, G7 end-to-end FUNSD
    c, p = W.synth_craft(0, True), W.synth_parseq(0)
    craft, parseq = pipeline.load_models(c, p)
    crops = np.random.default_rng(0).integers(0, 256, (8, 32, 128, 3), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "g5_parseq.npz"), logits=pipeline.parseq_logits(parseq, crops))
    craft_r, _ = pipeline.load_models(W.synth_craft(0, False), p)
    canvas = np.random.default_rng(1).integers(0, 256, (64, 96, 3), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "g6_craft.npz"), heat=pipeline.craft_heatmap(craft_r, canvas))
    from PIL import Image
    img = np.array(Image.open(os.path.join(ROOT, "tests", "data", "funsd_0001129658.png")).convert("RGB"))
    d = pipeline.image_to_data(craft, parseq, img, debug=True)
    out["funsd"] = d["result"]
    np.savez_compressed(os.path.join(HERE, "g7_funsd.npz"), det=d["det"], boxes=d["boxes"], crops=d["crops"][:6],
                        logits=d["logits"].astype(np.float32))
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("golden written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
