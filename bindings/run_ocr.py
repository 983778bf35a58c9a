#!/usr/bin/env python3
"""Counterpart of the reference's demo harness (bindings/run_ocr.py:85-107): open an image with PIL,
.convert("RGB"), hand the numpy array to pytuatara.image_to_data(image, weights_dir, outputs_dir), print the
result, and save an annotated copy (the reference's three panels: boxes on the page, text at the boxes, running text).  Drawing uses PIL only
(the reference draws with cv2 and opens a window; neither is needed for the results).

  python bindings/run_ocr.py [image] [weights_dir] [outputs_dir]
"""
from __future__ import annotations

import os
import sys

import numpy as np
from PIL import Image, ImageDraw

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.append(os.path.join(HERE, "..", "build", "bindings"))   # where the build puts pytuatara (reference: run_ocr.py:6)


def annotate(image: np.ndarray, result) -> Image.Image:
    """The reference's three panels side by side (run_ocr.py:10-82), drawn with PIL: the page with its boxes | each text at its
    box position | the texts as running text in reading order - sorted by (y1, x1) (:12), starting at (10, 30), wrapped at the
    page width, 10 px between words and lines (:20-25, :62-75)."""
    page = Image.fromarray(image).convert("RGB")
    w, h = page.size
    boxes = page.copy()
    panel = Image.new("RGB", page.size, "black")
    running = Image.new("RGB", page.size, "black")
    db, dp, dr = ImageDraw.Draw(boxes), ImageDraw.Draw(panel), ImageDraw.Draw(running)
    tx, ty, gap = 10, 30, 10
    for item in sorted(result, key=lambda it: (it["bbox"][1], it["bbox"][0])):
        x1, y1, x2, y2 = (int(v) for v in item["bbox"])
        text = item["text"]
        db.rectangle([x1, y1, x2, y2], outline=(0, 255, 0), width=2)
        dp.text((x1, y1), text, fill=(255, 0, 0))
        l, t, r, btm = dr.textbbox((0, 0), text or " ")
        tw, th = r - l, btm - t
        if tx + tw > w:
            tx = 10
            ty += th + gap
        dr.text((tx, ty - th), text, fill=(255, 0, 0))
        tx += tw + gap
    out = Image.new("RGB", (3 * w, h), "black")
    for k, im in enumerate((boxes, panel, running)):
        out.paste(im, (k * w, 0))
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    image_path = argv[0] if len(argv) > 0 else os.path.join(HERE, "..", "tests", "data", "funsd_0001129658.png")
    weights_dir = argv[1] if len(argv) > 1 else os.path.join(HERE, "..", "weights")
    outputs_dir = argv[2] if len(argv) > 2 else os.path.join(HERE, "..", "outputs")
    import pytuatara

    numpy_image = np.array(Image.open(image_path).convert("RGB"))
    result = pytuatara.image_to_data(numpy_image, weights_dir, outputs_dir)
    print(result)
    os.makedirs(outputs_dir, exist_ok=True)
    stem = os.path.splitext(os.path.basename(image_path))[0]
    annotate(numpy_image, result).save(os.path.join(outputs_dir, stem + "_annotated_with_ocr_results.png"))
    return result


if __name__ == "__main__":
    main()
