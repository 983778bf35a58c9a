"""GPU box tool: are two builds of the library the same arithmetic?  Heat maps (random CRAFT weights, four canvas sizes), recogniser logits and ids (400 crops, some
scaled so that planes near the f16 range occur) and whole-page results under TUATARA_LIB=<a> and <b>, each in its own process; compared bit for bit.
    python3 tools/ab_bits.py <lib a> <lib b>            (parent: starts the two dumps, compares)
    python3 tools/ab_bits.py --dump out.npz            (child)"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(path):
    from tuatara_amd import weights as W
    from tuatara_amd.engine import Engine
    d = tempfile.mkdtemp()
    W.export_craft(W.synth_craft(3, False), d); W.export_parseq(W.synth_parseq(0), d)
    eng = Engine(d)
    rng = np.random.default_rng(11)
    out = {}
    for i, hw in enumerate(((1024, 768), (512, 1024), (256, 192), (96, 160))):
        out[f"heat{i}"] = eng.craft_heatmap(rng.integers(0, 256, hw + (3,), dtype=np.uint8))
    crops = rng.integers(0, 256, (400, 32, 128, 3), dtype=np.uint8)
    crops[::7] //= 8                                                    # dark crops: another range of magnitudes
    lg, ids = eng.parseq_logits(crops)
    out["logits"], out["ids"] = lg, ids
    d2 = tempfile.mkdtemp(); W.make_synthetic_weights(d2, seed=0, structured=True)
    eng2 = Engine(d2)
    lg2, ids2 = eng2.parseq_logits(crops[:64])
    out["logits_structured"], out["ids_structured"] = lg2, ids2
    page = rng.integers(0, 256, (768, 1024, 3), dtype=np.uint8)
    page[200:260, 100:700] = 0; page[400:440, 300:900] = 255
    res = eng2.image_to_data(page)
    out["page_boxes"] = np.array([r["bbox"] for r in res], dtype=np.float64).reshape(-1, 4)
    out["page_ids"] = np.array([r["ids"] for r in res], dtype=np.int64).reshape(-1, 26)
    np.savez(path, **out)


if __name__ == "__main__":
    if sys.argv[1] == "--dump":
        dump(sys.argv[2]); sys.exit(0)
    outs = []
    for lib in sys.argv[1:3]:
        f = tempfile.mktemp(suffix=".npz")
        env = dict(os.environ, TUATARA_LIB=os.path.abspath(lib))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--dump", f], env=env)
        outs.append(np.load(f))
    a, b = outs
    bad = 0
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k], b[k])
        extra = "" if same or a[k].dtype.kind in "US" or a[k].shape != b[k].shape else f"  {int((a[k] != b[k]).sum())} values differ, max {float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()):.3e}"
        print(f"{k:20s} {str(a[k].shape):22s} {'identical' if same else 'DIFFERENT'}{extra}")
        bad += not same
    print("ab_bits:", "every output identical" if not bad else f"{bad} outputs differ")
    sys.exit(1 if bad else 0)
