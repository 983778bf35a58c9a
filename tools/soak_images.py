"""GPU box tool: the list form (ttr_images_to_data) under churn, default precision with the recogniser on its own stream: R rounds over a few hundred images of
many sizes (seeded), shuffled differently every round, batches of a different size every round - every image's result must equal its first round's, whatever
its neighbours and its place in the list; the streamed pipeline, both streams, the staging slots and the workspaces' growth see every order.
   python3 tools/soak_images.py [images=240] [rounds=6]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import synth, weights as W
from tuatara_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = tempfile.mkdtemp(); W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d)
rng = np.random.default_rng(0)
sizes = [(1024, 768), (768, 1024), (1000, 754), (512, 384), (763, 607), (206, 275), (664, 1245), (96, 160), (1536, 1152), (333, 517)]
base = {s: [synth.synthetic_page(900 + 7 * k + i, max(s[0], 64), max(s[1], 64), n_words=int(rng.integers(3, 30))) for i in range(3)] for k, s in enumerate(sizes)}
imgs = []
for i in range(n):
    s = sizes[int(rng.integers(0, len(sizes)))]
    im = base[s][int(rng.integers(0, 3))].copy()
    y, x = int(rng.integers(0, max(1, s[0] - 20))), int(rng.integers(0, max(1, s[1] - 60)))
    im[y:y + 8, x:x + 50] = 0                                   # every image differs
    imgs.append(im)
def key(r): return [(tuple(x["bbox"]), x["text"]) for x in r]
t0 = time.time()
ref = [key(r) for r in eng.images_to_data(imgs)]
print(f"{n} images of {len(sizes)} sizes, {sum(len(r) for r in ref)} boxes, first round {time.time() - t0:.1f} s", flush=True)
bad = 0
for it in range(rounds):
    perm = rng.permutation(n)
    assert eng.set_tuning(b"images_batch", int(rng.choice([1, 2, 3, 5, 8, 16, 32]))) == 0
    got = eng.images_to_data([imgs[i] for i in perm])
    wrong = [int(perm[k]) for k in range(n) if key(got[k]) != ref[perm[k]]]
    if wrong:
        bad += 1
        print("round", it, "differs on images", wrong[:10], flush=True)
# single calls agree too (a sample)
for i in rng.choice(n, 12, replace=False):
    if key(eng.image_to_data(imgs[i])) != ref[i]:
        bad += 1; print("single call differs on image", int(i))
print(f"soak_images: {rounds} shuffled rounds, {bad} differences, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
