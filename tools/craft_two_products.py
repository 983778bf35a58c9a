"""GPU box experiment: CRAFT on TWO products per value - x0 w0 + x1 w0 / 2^11, the weights as ONE f16 plane - against the engine's three
(x0 w1 as well).  The two-product arithmetic is reproduced on the three-product kernels by zeroing the w1 planes of every layer with >= 64 input channels
(TUATARA_CRAFT_PRODUCTS=2, engine.cpp: load_craft); the time it would save is known without building it: a third of those layers' MFMAs, ~15 ms of the
47.5 ms detector pass.  The gate: every box np.array_equal to the three-product engine's and every string identical, on FUNSD and on the benchmark's page
stream.  Prints the heat-map error, the threshold flips and the differing boxes / strings either way.

  python3 tools/craft_two_products.py [pages]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import synth, weights as W
from tuatara_amd.engine import DeviceBuffer, Engine

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = tempfile.mkdtemp()
W.make_synthetic_weights(d, seed=0, structured=True)
os.environ.pop("TUATARA_CRAFT_PRODUCTS", None)
e3 = Engine(d)
os.environ["TUATARA_CRAFT_PRODUCTS"] = "2"
e2 = Engine(d)
os.environ.pop("TUATARA_CRAFT_PRODUCTS", None)


def norm(h):
    return (h - h.min()) / (h.max() - h.min())


def compare(name, pages):
    tot = dict(pages=0, boxes=0, box_diff=0, count_diff=0, str_diff=0, flips_text=0, flips_link=0, flips_07=0, max_dheat=0.0)
    for pg in pages:
        r3, r2 = e3.image_to_data(pg), e2.image_to_data(pg)
        tot["pages"] += 1
        tot["boxes"] += len(r3)
        if len(r3) != len(r2):
            tot["count_diff"] += 1
        b3, b2 = {tuple(x["bbox"]): x["text"] for x in r3}, {tuple(x["bbox"]): x["text"] for x in r2}
        tot["box_diff"] += len(set(b3) ^ set(b2)) // 2 + abs(len(b3) - len(b2)) // 2 if set(b3) != set(b2) else 0
        tot["str_diff"] += sum(1 for k in b3 if k in b2 and b3[k] != b2[k])
        canvas, _ = e3.resize_canvas(pg)
        h3, h2 = e3.craft_heatmap(canvas), e2.craft_heatmap(canvas)
        tot["max_dheat"] = max(tot["max_dheat"], float(np.abs(h3 - h2).max()))
        t3, t2, l3, l2 = norm(h3[..., 0]), norm(h2[..., 0]), norm(h3[..., 1]), norm(h2[..., 1])
        tot["flips_text"] += int(((t3 > 0.4) != (t2 > 0.4)).sum())
        tot["flips_link"] += int(((l3 > 0.4) != (l2 > 0.4)).sum())
        tot["flips_07"] += int(((t3 >= 0.7) != (t2 >= 0.7)).sum())
    print(name, tot, flush=True)
    return tot


from PIL import Image
funsd = np.array(Image.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "data", "funsd_0001129658.png")).convert("RGB"))
a = compare("FUNSD", [funsd])
b = compare(f"stream of {n_pages} synthetic 1024x768 pages (40 words, jitter4)", [synth.synthetic_page(s, 1024, 768, n_words=40) for s in range(n_pages)])
ok = all(t["box_diff"] == 0 and t["count_diff"] == 0 and t["str_diff"] == 0 for t in (a, b))
print("GATE:", "PASS - worth building" if ok else "FAIL - two products change boxes / strings: not built")
