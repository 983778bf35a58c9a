#!/usr/bin/env python3
"""Counterpart of the reference's demo harness (bindings/run_ocr.py:85-107): open an image with PIL,
.convert("RGB"), hand the numpy array to pytuatara.image_to_data(image, weights_dir, outputs_dir), print the
result, and save an annotated copy (boxes on the page beside the recognised text).  Drawing uses PIL only
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
    """Page with boxes | white panel with each text drawn at its box position, reading order (y1, x1)."""
    page = Image.fromarray(image).convert("RGB")
    boxes = page.copy()
    panel = Image.new("RGB", page.size, "white")
    db, dp = ImageDraw.Draw(boxes), ImageDraw.Draw(panel)
    for item in sorted(result, key=lambda it: (it["bbox"][1], it["bbox"][0])):
        x1, y1, x2, y2 = item["bbox"]
        db.rectangle([x1, y1, x2, y2], outline=(0, 160, 0), width=2)
        dp.rectangle([x1, y1, x2, y2], outline=(200, 200, 200), width=1)
        dp.text((x1 + 2, y1 + 1), item["text"], fill=(0, 0, 0))
    out = Image.new("RGB", (page.size[0] * 2, page.size[1]), "white")
    out.paste(boxes, (0, 0))
    out.paste(panel, (page.size[0], 0))
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
