"""-m gpu: the integer / byte stages through the C ABI, bit-exact against the oracle:
page resize+pad+swap, union-find CCL + box extraction, crop-batch packer."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f16x4", "f32"])
def eng_post(request, eng_x4, eng_f32):
    """The integer / byte stages do not depend on the precision, but the SHIPPED engine object is the f16x4 one: every test here runs on it, and on the
    fp32 engine beside it."""
    return eng_x4 if request.param == "f16x4" else eng_f32


@pytest.mark.parametrize("hw", [(1000, 754), (763, 607), (206, 275), (664, 1245), (2048, 1536), (2000, 1500), (33, 700)])
def test_resize_canvas_bit_exact(eng_post, hw):
    from oracle import post
    h, w = hw
    img = np.random.default_rng(h * 7 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    got, ratio = eng_post.resize_canvas(img)
    ref, rratio = post.resize_aspect_ratio(np.ascontiguousarray(img[:, :, ::-1]))   # swap (:349) then resize+pad
    assert got.shape == ref.shape and ratio == rratio
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ccl_boxes_match_oracle_and_golden(eng_post, seed):
    from oracle import post
    from tests.golden.make_golden import synthetic_heatmap
    heat = synthetic_heatmap(seed)
    got = eng_post.ccl_boxes(heat)
    ref, _, _ = post.get_detected_boxes(heat[..., 0], heat[..., 1])
    g = np.load(os.path.join(GOLDEN, "g4_boxes.npz"))[f"rects_{seed}"]
    assert got.shape == ref.shape == g.shape and len(got) >= 15
    assert np.array_equal(got, ref)           # same order (component label order), same float32 rects
    assert np.allclose(got, g, atol=1e-4)


def test_ccl_full_size_and_edge_cases(eng_post):
    from oracle import post
    from tests.golden.make_golden import synthetic_heatmap
    big = synthetic_heatmap(7, 512, 384)                                  # BASELINE config 3 heat-map size
    got = eng_post.ccl_boxes(big)
    ref, _, _ = post.get_detected_boxes(big[..., 0], big[..., 1])
    assert len(ref) >= 60 and np.array_equal(got, ref)
    # one giant component (noise-only link map floods the page) and an empty page
    rng = np.random.default_rng(0)
    flood = np.stack([synthetic_heatmap(3)[..., 0], rng.normal(0, 0.01, (256, 192)).astype(np.float32)], -1)
    ref, _, _ = post.get_detected_boxes(flood[..., 0], flood[..., 1])
    assert np.array_equal(eng_post.ccl_boxes(flood), ref)
    ramp = np.zeros((64, 64, 2), np.float32)
    ramp[..., 0] = np.linspace(0, 0.2, 64)[None, :]
    ramp[..., 1] = np.linspace(0, 0.1, 64)[:, None]
    ref, _, _ = post.get_detected_boxes(ramp[..., 0], ramp[..., 1])
    assert np.array_equal(eng_post.ccl_boxes(ramp), ref)


def test_ccl_serpentine_union_find_stress(eng_post):
    """A long snake forces deep union-find chains across workgroups; labels must still merge into one."""
    from oracle import post
    H, W = 128, 160
    t = np.zeros((H, W), np.float32)
    for y in range(2, H - 2, 4):
        t[y, 2:W - 2] = 1.0
        t[y:y + 4, (W - 3) if (y // 4) % 2 == 0 else 2] = 1.0
    heat = np.stack([t, np.zeros_like(t)], -1)
    heat[0, 0, 1] = 1.0
    got = eng_post.ccl_boxes(heat)
    ref, labels, _ = post.get_detected_boxes(heat[..., 0], heat[..., 1])
    assert labels.max() == 2 and len(ref) == 1
    assert np.array_equal(got, ref)


def test_pack_crops_bit_exact(eng_post, funsd):
    from oracle import post
    g = np.load(os.path.join(GOLDEN, "g7_funsd.npz"))
    det = g["det"]
    crops, boxes = eng_post.pack_crops(funsd, det, 1.0)
    swapped = np.ascontiguousarray(funsd[:, :, ::-1])
    ref_boxes = post.adjust_result_coordinates(det, 1.0, 1.0)
    assert np.array_equal(boxes, ref_boxes)
    for i, b in enumerate(ref_boxes):
        assert np.array_equal(crops[i], post.crop_resize(swapped, b, clamp=True)), i
    assert np.array_equal(crops[:6], g["crops"])


def test_pack_crops_colour_and_ragged_sizes(eng_post):
    """Colour image (channel order matters), up- and down-scaling crops, boxes poking outside."""
    from oracle import post
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    rects = np.array([[50, 40, 60, 20, 0], [100, 75, 16, 6, -90], [198, 10, 30, 30, 30], [5, 140, 40, 12, 5],
                      [120, 100, 238, 198, 0], [60, 60, 3, 2, 0]], np.float32)      # heat-map units (x2 -> image)
    crops, boxes = eng_post.pack_crops(img, rects, 1.0)
    swapped = np.ascontiguousarray(img[:, :, ::-1])
    for i, b in enumerate(post.adjust_result_coordinates(rects, 1.0, 1.0)):
        ref = post.crop_resize(swapped, b, clamp=True)
        assert ref is not None and np.array_equal(crops[i], ref), i


@pytest.mark.parametrize("hw", [(96, 112), (384, 304), (128, 64), (64, 448)])
def test_ccl_statistics_paths(eng_post, hw):
    """Component statistics are folded per wave where a wave's 64 pixels share a row (W a multiple of 64) and accumulated per pixel
    otherwise; both must give the oracle's boxes (area / bbox / text maximum feed the candidate filter and the calipers)."""
    from oracle import post
    from tests.golden.make_golden import synthetic_heatmap
    heat = synthetic_heatmap(11 + hw[1], hw[0], hw[1])
    got = eng_post.ccl_boxes(heat)
    ref, _, _ = post.get_detected_boxes(heat[..., 0], heat[..., 1])
    assert len(ref) >= 3 and np.array_equal(got, ref)


def test_gpu_side_min_area_rect_equals_the_hosts(eng_post):
    """get_detected_boxes' per-component tail on the GPU (post_ops.hip: ccl_rects_kernel, one lane per candidate running geometry.cpp's arithmetic; the default,
    tuning key "gpu_calipers") against the host's calipers, and against the path that falls back to the host when the hulls' scratch pool is too small
    (gpu_calipers = 2: a 2 KB pool): the same float32 rectangles in the same order, bit for bit - on the synthetic heat maps of the golden set (40 anisotropic
    blobs, bridges, edge-touching and tiny components) and on a full-size map with a page-wide component."""
    from oracle import post
    from tests.golden.make_golden import synthetic_heatmap
    maps = [synthetic_heatmap(seed) for seed in (0, 1, 2, 3)]
    big = np.zeros((512, 384, 2), np.float32)
    big[40:470, 30:350, 0] = 1.0                      # one component of 430 rows: ~900 hull candidates
    big[10:14, 10:60, 0] = 0.9
    maps.append(big)
    for hm in maps:
        ref = eng_post.ccl_boxes(hm)
        outs = []
        for k in (0, 2):
            assert eng_post.set_tuning(b"gpu_calipers", k) == 0
            try:
                outs.append(eng_post.ccl_boxes(hm))
            finally:
                eng_post.set_tuning(b"gpu_calipers", 1)
        assert len(ref) > 0 and np.array_equal(ref, outs[0]) and np.array_equal(ref, outs[1])
        oracle_rects, _, _ = post.get_detected_boxes(hm[..., 0], hm[..., 1])
        assert np.array_equal(ref, oracle_rects)
