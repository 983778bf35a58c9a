"""Synthetic workload of BASELINE.json config 5: 1024x768 pages, white background,
~40 random alphanumeric words drawn with PIL's built-in bitmap font at seeded positions."""
from __future__ import annotations

import numpy as np

ALNUM = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"


def synthetic_page(seed: int, h: int = 1024, w: int = 768, n_words: int = 40, scale: int = 2, layout: str = "jitter4") -> np.ndarray:
    """u8 [h,w,3].  layout "jitter4" (the parity tests' pages): words sit on a 4-column jittered grid so they never touch.  layout "cells5x8"
    (the benchmark's pages): one word inside each cell of the 5 x 8 grid whose 150 x 40 px boxes `bench_grid_boxes` hands to the recogniser
    (bench.py --boxes=grid40), so that the 40 crops of a page frame text and decode to strings of the words' lengths.  Glyphs are the PIL
    default font magnified `scale` x (nearest) so strokes survive the detector's stride-2 heat map; same words per seed in both layouts."""
    from PIL import Image, ImageDraw, ImageFont

    rng = np.random.default_rng(seed)
    font = ImageFont.load_default()
    page = Image.new("L", (w, h), 255)
    if layout == "cells5x8":
        for k in range(min(n_words, 40)):
            r, c = divmod(k, 5)
            word = "".join(rng.choice(list(ALNUM), size=int(rng.integers(3, 11))))
            tile = Image.new("L", (70, 14), 255)
            ImageDraw.Draw(tile).text((1, 1), word, fill=0, font=font)
            bbox = Image.eval(tile, lambda v: 255 - v).getbbox()
            # the word fills its box (the recogniser's crop is the box stretched to 128 x 32 with no regard for the aspect ratio,
            # tuatara.cpp:440: a short word left at font scale would leave most of the crop blank paper)
            tw, th = int(rng.integers(136, 145)), int(rng.integers(30, 35))
            tile = tile.crop((bbox[0], bbox[1], bbox[2] + 1, bbox[3] + 1)).resize((tw, th), Image.NEAREST)
            cx, cy = (c + 0.5) * w / 5.0, (r + 0.5) * h / 8.0                      # the grid box: 150 x 40 px about (cx, cy)
            x0, x1 = int(cx - 75) + 2, int(cx + 75) - 2 - tw
            y0, y1 = int(cy - 20) + 2, int(cy + 20) - 2 - th
            x = int(rng.integers(x0, max(x0 + 1, x1 + 1)))
            y = int(rng.integers(y0, max(y0 + 1, y1 + 1)))
            page.paste(tile, (x, y))
        a = np.asarray(page, dtype=np.uint8)
        return np.ascontiguousarray(np.repeat(a[:, :, None], 3, 2))
    if layout != "jitter4":
        raise ValueError("layout must be 'jitter4' or 'cells5x8'")
    cols, rows = 4, (n_words + 3) // 4
    cw, ch = w // cols, h // rows
    k = 0
    for r in range(rows):
        for c in range(cols):
            if k >= n_words:
                break
            word = "".join(rng.choice(list(ALNUM), size=int(rng.integers(3, 11))))
            tile = Image.new("L", (70, 14), 255)
            ImageDraw.Draw(tile).text((1, 1), word, fill=0, font=font)
            bbox = Image.eval(tile, lambda v: 255 - v).getbbox()
            tile = tile.crop((0, 0, bbox[2] + 1, 14)).resize(((bbox[2] + 1) * scale, 14 * scale), Image.NEAREST)
            x = c * cw + int(rng.integers(4, max(5, cw - tile.size[0] - 4)))
            y = r * ch + int(rng.integers(4, max(5, ch - tile.size[1] - 4)))
            page.paste(tile, (x, y))
            k += 1
    a = np.asarray(page, dtype=np.uint8)
    return np.ascontiguousarray(np.repeat(a[:, :, None], 3, 2))
