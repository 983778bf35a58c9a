"""Synthetic workload of BASELINE.json config 5: 1024x768 pages, white background,
~40 random alphanumeric words drawn with PIL's built-in bitmap font at seeded positions."""
from __future__ import annotations

import numpy as np

ALNUM = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"


def synthetic_page(seed: int, h: int = 1024, w: int = 768, n_words: int = 40, scale: int = 2) -> np.ndarray:
    """u8 [h,w,3].  Words sit on an 8 x 5 jittered grid so they never touch; glyphs are the PIL
    default font magnified `scale` x (nearest) so strokes survive the detector's stride-2 heat map."""
    from PIL import Image, ImageDraw, ImageFont

    rng = np.random.default_rng(seed)
    font = ImageFont.load_default()
    page = Image.new("L", (w, h), 255)
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
