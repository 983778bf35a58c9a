"""CPU suite: the oracle against its golden vectors, independent numpy/scipy cross-checks of the
OpenCV restatement, and the reference quirks recorded in SURVEY.md section 8a (N1, N4, N5)."""
import json
import os

import numpy as np
import pytest
from scipy import ndimage as ndi

from tests.conftest import GOLDEN
from oracle import post

G = json.load(open(os.path.join(GOLDEN, "golden.json")))


# ---------------------------------------------------------------- tokenizer (tuatara.cpp:25-117)
def test_tokenizer_table_quirks():
    itos, eos, bos, pad = post.tokenizer_table()
    assert len(itos) == 98                      # 95-char charset ("\\'" is two chars) + EOS + BOS + PAD
    assert (eos, bos, pad) == (88, 96, 97)      # std::map: the LAST index of a duplicate char wins
    assert itos[0] == "]" and itos[96] == "[" and itos[97] == "P"
    assert itos[1:11] == "0123456789" and itos[11:37] == "abcdefghijklmnopqrstuvwxyz"
    # ids >= 69 are shifted by one relative to upstream PARSeq (the annotated PNG shows ':'->'/', '-'->',', ...)
    shifted = {69: "\\", 70: "'", 75: ",", 76: "-", 77: ".", 78: "/", 79: ":", 88: "]", 94: "}"}
    for i, ch in shifted.items():
        assert itos[i] == ch
    assert itos == G["tokenizer"]["itos"]


def test_tokenizer_decode_golden():
    for ids, s in G["tokenizer"]["cases"]:
        assert post.decode_ids(ids) == s
    assert post.decode_ids([12, 13, 0, 14]) == "bc"        # cut at the first EOS char (id 0)
    assert post.decode_ids([12, 88, 13]) == "bc"           # every id 88 is filtered out, not a terminator
    assert post.decode_ids([]) == ""


def test_decode_logits_argmax_first_max():
    lg = np.zeros((1, 3, 95), np.float32)
    lg[0, 0, 5] = lg[0, 0, 7] = 3.0   # tie -> first index
    lg[0, 1, 0] = 1.0                 # EOS
    s, ids = post.decode_logits(lg)
    assert ids[0].tolist() == [5, 0, 0] and s == ["4"]


# ---------------------------------------------------------------- resize_aspect_ratio (tuatara.cpp:206-234)
def test_resize_dims_golden_and_survey_table():
    for h, w, exp in G["resize_dims"]:
        got = post.resize_aspect_ratio_dims(h, w)
        assert list(got[:4]) == exp[:4] and abs(got[4] - exp[4]) < 1e-7
    assert post.resize_aspect_ratio_dims(1000, 754)[:4] == (1000, 754, 1024, 768)
    th, tw, h32, w32, r = post.resize_aspect_ratio_dims(664, 1245)
    assert (th, tw, h32, w32) == (546, 1024, 576, 1024) and abs(r - 0.82249) < 1e-5
    assert post.resize_aspect_ratio_dims(2000, 1500)[:4] == (1024, 768, 1024, 768)


def test_resize_linear_matches_float_bilinear_within_1lsb():
    rng = np.random.default_rng(0)
    for (sh, sw, dh, dw) in [(40, 150, 32, 128), (11, 37, 32, 128), (90, 300, 32, 128), (100, 80, 73, 61)]:
        img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
        got = post.resize_linear(img, dh, dw).astype(np.float64)
        ys = np.clip((np.arange(dh) + 0.5) * sh / dh - 0.5, 0, sh - 1)
        xs = np.clip((np.arange(dw) + 0.5) * sw / dw - 0.5, 0, sw - 1)
        y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
        y1, x1 = np.minimum(y0 + 1, sh - 1), np.minimum(x0 + 1, sw - 1)
        fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
        f = img.astype(np.float64)
        ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx) * (1 - fy) + (f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx) * fy
        assert np.abs(got - ref).max() <= 1.0 + 1e-9


def test_resize_identity_and_exact_half():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    assert np.array_equal(post.resize_linear(img, 20, 30), img)
    half = post.resize_linear(img, 10, 15)      # OpenCV turns an exact 2x2 INTER_LINEAR decimation into INTER_AREA
    ref = (img.reshape(10, 2, 15, 2, 3).astype(np.int32).sum((1, 3)) + 2) >> 2
    assert np.array_equal(half, ref.astype(np.uint8))


# ---------------------------------------------------------------- connected components (tuatara.cpp:142)
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_ccl_matches_scipy_label_order_and_stats(seed):
    rng = np.random.default_rng(seed)
    img = (ndi.uniform_filter(rng.random((70, 90)), 3) > (0.5 + 0.03 * seed)).astype(np.uint8)
    n, labels, stats = post.connected_components4(img)
    ref, nref = ndi.label(img, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    assert n == nref + 1
    assert np.array_equal(labels, ref)          # raster-first-pixel numbering (OpenCV's)
    for k, sl in enumerate(ndi.find_objects(ref), 1):
        assert stats[k].tolist() == [sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start, int((ref == k).sum())]


def test_ccl_edge_cases():
    assert post.connected_components4(np.zeros((5, 7), np.uint8))[0] == 1
    n, labels, stats = post.connected_components4(np.ones((5, 7), np.uint8))
    assert n == 2 and stats[1].tolist() == [0, 0, 7, 5, 35]
    diag = np.eye(6, dtype=np.uint8)            # 4-connectivity: diagonal pixels are separate components
    assert post.connected_components4(diag)[0] == 7


# ---------------------------------------------------------------- minAreaRect (tuatara.cpp:179)
def _brute_min_area(pts):
    best = None
    for ang in np.linspace(0, np.pi / 2, 9001):
        c, s = np.cos(ang), np.sin(ang)
        u, v = pts @ np.array([c, s]), pts @ np.array([-s, c])
        a = (u.max() - u.min()) * (v.max() - v.min())
        best = a if best is None or a < best else best
    return best


@pytest.mark.parametrize("seed", range(6))
def test_min_area_rect_is_minimal_and_encloses(seed):
    rng = np.random.default_rng(seed)
    th = rng.uniform(0, np.pi)
    pts = rng.normal(size=(60, 2)) * [20, 4]
    pts = np.round(pts @ np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]]) + 100)
    r = post.min_area_rect(pts)                       # OpenCV-style float32 rotating calipers
    e = post.min_area_rect(pts, exhaustive=True)      # independent: every hull edge, double precision
    assert abs(r[2] * r[3] - e[2] * e[3]) <= 1e-4 * e[2] * e[3]
    assert np.abs(np.sort(post.rect_points(r), 0) - np.sort(post.rect_points(e), 0)).max() < 2e-2
    assert r[2] * r[3] <= _brute_min_area(pts) * (1 + 1e-4)
    corners = post.rect_points(r).astype(np.float64)
    # all points inside the rectangle (half-plane test on the 4 edges)
    ctr = corners.mean(0)
    for i in range(4):
        a, b = corners[i], corners[(i + 1) % 4]
        nrm = np.array([-(b - a)[1], (b - a)[0]])
        if nrm @ (ctr - a) < 0:
            nrm = -nrm
        assert ((pts - a) @ nrm >= -1e-3 * np.linalg.norm(nrm)).all()


def test_min_area_rect_axis_aligned_exact_and_degenerate():
    ys, xs = np.mgrid[10:15, 20:61]
    r = post.min_area_rect(np.stack([xs.ravel(), ys.ravel()], 1))
    pts = post.rect_points(r)
    assert sorted(map(tuple, pts.tolist())) == [(20.0, 10.0), (20.0, 14.0), (60.0, 10.0), (60.0, 14.0)]
    assert post.bounding_rect(r) == (20, 10, 41, 5)
    assert post.tesseract_bbox(r) == [20.0, 10.0, 60.0, 14.0]
    line = post.min_area_rect([[0, 0], [10, 0], [5, 0]])       # collinear -> n == 2 branch
    assert line[2] == 10 and line[3] == 0
    assert post.min_area_rect([[3, 4]])[:2].tolist() == [3.0, 4.0]


# ---------------------------------------------------------------- get_detected_boxes (tuatara.cpp:119-204)
def _boxes_numpy(text, link, text_threshold=0.7, link_threshold=0.4, low_text=0.4):
    """Independent numpy/scipy walk through tuatara.cpp:119-204 (full-image masks, scipy dilation)."""
    H, W = text.shape
    tn = (text - text.min()) / (text.max() - text.min())
    ln = (link - link.min()) / (link.max() - link.min())
    ts, ls = tn > np.float32(low_text), ln > np.float32(link_threshold)
    lab, n = ndi.label(ts | ls, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    out = []
    for k, sl in enumerate(ndi.find_objects(lab), 1):
        m = lab == k
        size = int(m.sum())
        if size < 10 or tn[m].max() < np.float32(text_threshold):
            continue
        seg = m & ~(ls & ~ts)
        x, y, w, h = sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start
        niter = int(np.sqrt(size * min(w, h) // (w * h) * 2))       # N4: integer arithmetic inside the sqrt
        sx, sy, ex, ey = max(0, x - niter), max(0, y - niter), min(W, x + w + niter + 1), min(H, y + h + niter + 1)
        ks = 1 + niter
        org = -1 if ks % 2 == 0 else 0                               # N5: OpenCV anchor k/2 for an even kernel
        dil = ndi.binary_dilation(seg, structure=np.ones((ks, ks), bool), origin=(org, org))
        seg2 = seg.copy()
        seg2[sy:ey, sx:ex] = dil[sy:ey, sx:ex]
        ys, xs = np.nonzero(seg2)
        out.append(post.min_area_rect(np.stack([xs, ys], 1)))
    return np.array(out, np.float32).reshape(-1, 5)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_get_detected_boxes_golden_and_numpy(seed):
    from tests.golden.make_golden import synthetic_heatmap
    heat = synthetic_heatmap(seed)
    rects, labels, tn = post.get_detected_boxes(heat[..., 0], heat[..., 1])
    g = np.load(os.path.join(GOLDEN, "g4_boxes.npz"))
    assert np.allclose(rects, g[f"rects_{seed}"], atol=1e-4)
    assert labels.max() == int(g[f"nlabels_{seed}"][0])
    ref = _boxes_numpy(heat[..., 0], heat[..., 1])
    assert ref.shape == rects.shape and len(rects) >= 15
    assert np.allclose(rects, ref, atol=1e-4)


def test_niter_integer_arithmetic():
    """N4: size*min(w,h)/(w*h)*2 truncates before the *2 (upstream CRAFT uses float math)."""
    # a 30x6 solid word: 180*6/180*2 = 12 -> niter 3 ; float math would give sqrt(12)=3.46 -> 3 as well,
    # a sparse 40x10 component of 150 px: 150*10/400 = 3 (3.75 truncated) *2 = 6 -> 2 ; float: sqrt(7.5) -> 2
    # a 50x4 component of 130 px: 130*4/200 = 2 (2.6) *2 = 4 -> 2 ; float math: sqrt(5.2) = 2.28 -> 2
    # a 31x3 component of 61 px: 61*3/93 = 1 (1.97) * 2 = 2 -> 1 ; float math sqrt(3.93) = 1.98 -> 1
    # a 20x7 component of 139 px: 139*7/140 = 6 (6.95) * 2 = 12 -> 3 ; float math sqrt(13.9) = 3.7 -> 3
    # a 9x8 component of 71 px: 71*8/72 = 7 (7.9) *2 = 14 -> 3 ; float: sqrt(15.8) = 3.97 -> 3
    # a 10x10 component of 99 px: 99*10/100 = 9 *2 = 18 -> 4 ; float: sqrt(19.8) = 4.45 -> 4
    # a 16x16 component of 255 px: 255*16/256 = 15 *2 = 30 -> 5 ; float: sqrt(31.9) = 5.6 -> 5
    # a 5x5 component of 24 px: 24*5/25 = 4 *2 = 8 -> 2 ; float: sqrt(9.6)=3.1 -> 3  <-- differs
    t = np.zeros((40, 40), np.float32)
    t[10:15, 10:15] = 1.0
    t[10, 10] = 0.0          # 24 px in a 5x5 box
    l = np.zeros_like(t)
    l[0, 0] = 1.0            # keeps link normalisation finite; single pixel is its own tiny component
    rects, _, _ = post.get_detected_boxes(t, l)
    assert len(rects) == 1
    # niter = 2 -> kernel 3x3 -> the 5x5 box grows by exactly 1 on every side (7x7 => extent 6), not by 3/2
    assert sorted([round(float(rects[0][2]), 3), round(float(rects[0][3]), 3)]) == [6.0, 6.0]


# ---------------------------------------------------------------- models vs golden
def test_parseq_oracle_golden(oracle_models):
    from oracle import pipeline
    crops = np.random.default_rng(0).integers(0, 256, (8, 32, 128, 3), dtype=np.uint8)
    lg = pipeline.parseq_logits(oracle_models[1], crops)
    g = np.load(os.path.join(GOLDEN, "g5_parseq.npz"))["logits"]
    assert np.abs(lg - g).max() < 2e-3          # torch CPU kernels differ slightly across hosts/thread counts
    assert np.array_equal(lg.argmax(-1), g.argmax(-1))


def test_parseq_early_exit_invariance(oracle_models):
    """The upstream data-dependent break does not change the refined logits, so the
    engine's fixed-trip-count decoder is parity-safe."""
    import torch
    crops = np.random.default_rng(5).integers(0, 256, (4, 32, 128, 3), dtype=np.uint8)
    x = torch.from_numpy(crops).permute(0, 3, 1, 2).float().div(255.0)
    a = oracle_models[1](x, early_exit=True).numpy()
    b = oracle_models[1](x, early_exit=False).numpy()
    assert a.shape == b.shape == (4, 26, 95)
    from tests.parity_rules import upto_eos
    mask = np.arange(26)[None, :] < upto_eos(b.argmax(-1))[:, None]       # what the reference's string cut keeps (tuatara.cpp:497-502)
    assert np.array_equal(a.argmax(-1)[mask], b.argmax(-1)[mask])
    assert np.abs(a - b)[mask].max() < 1e-3 and np.abs(a - b).max() < 5e-3  # fp32 summation order of the shorter batch shapes (logits reach ~30)


def test_craft_oracle_golden(weights_random):
    from oracle import pipeline
    craft, _ = pipeline.load_models(weights_random["craft"], weights_random["parseq"])
    canvas = np.random.default_rng(1).integers(0, 256, (64, 96, 3), dtype=np.uint8)
    heat = pipeline.craft_heatmap(craft, canvas)
    g = np.load(os.path.join(GOLDEN, "g6_craft.npz"))["heat"]
    assert heat.shape == (32, 48, 2)
    assert np.abs(heat - g).max() < 1e-3


def test_funsd_oracle_end_to_end_golden(oracle_models, funsd):
    """BASELINE config 1 on the CPU oracle: the FUNSD page decodes to the committed list."""
    from oracle import pipeline
    assert funsd.shape == (1000, 754, 3)
    d = pipeline.image_to_data(oracle_models[0], oracle_models[1], funsd, debug=True)
    g = np.load(os.path.join(GOLDEN, "g7_funsd.npz"))
    assert d["heat"].shape == (512, 384, 2)
    assert np.allclose(d["det"], g["det"], atol=1e-2) and np.allclose(d["boxes"], g["boxes"], atol=2e-2)
    assert np.array_equal(d["crops"][:6], g["crops"])
    assert [r["bbox"] for r in d["result"]] == [r["bbox"] for r in G["funsd"]]
    same = sum(a["text"] == b["text"] for a, b in zip(d["result"], G["funsd"]))
    assert same >= len(G["funsd"]) - 2          # torch CPU rounding may flip a near-tie on another host
