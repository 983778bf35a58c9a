"""CPU suite: the C-ABI library loads and exports every symbol include/*.h declares; the drop-in
surfaces (tuatara.h shim, pytuatara) keep the reference's error conventions; the engine's host
geometry agrees with the oracle.  No GPU compute here."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from tests.conftest import ROOT, has_gpu


@pytest.fixture(scope="module")
def built():
    from tuatara_amd import build
    build.build_all()
    return build


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "tuatara_hip.h")).read() + open(os.path.join(ROOT, "include", "tuatara_hip_debug.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ttr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    from tuatara_amd import engine
    lib = ctypes.CDLL(engine.lib_path())
    decl = _declared_symbols()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert sorted(n for n, _, _ in engine.SYMBOLS) == decl     # the python binding covers the whole ABI
    assert b"gfx950" in engine.load().ttr_version()


def test_config_defaults_are_the_reference_constants(built):
    from tuatara_amd import engine
    cfg = engine.Config()
    engine.load().ttr_config_default(ctypes.byref(cfg))
    assert (cfg.canvas_size, cfg.min_area) == (1024, 10)                       # tuatara.cpp:352, :148
    assert abs(cfg.mag_ratio - 1.0) < 1e-9
    assert np.allclose([cfg.text_threshold, cfg.link_threshold, cfg.low_text], [0.7, 0.4, 0.4])  # :397-399


def test_engine_fails_loudly_without_gpu_or_weights(built, tmp_path):
    from tuatara_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(str(tmp_path / "no_such_dir"))
    if not has_gpu():
        from tuatara_amd import weights as W
        W.make_synthetic_weights(str(tmp_path), 0)
        with pytest.raises(EngineError, match="no HIP device|HIP error"):
            Engine(str(tmp_path))


def test_pytuatara_surface_and_error_conventions(built, capfd):
    sys.path.insert(0, os.path.join(ROOT, "build", "bindings"))     # where the reference's run_ocr.py looks (run_ocr.py:6)
    import pytuatara
    img = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(RuntimeError, match="Input array should have 3 dimensions"):   # python.cpp:15-17
        pytuatara.image_to_data(np.zeros((8, 8), np.uint8), "w", "o")
    assert pytuatara.image_to_data(image=img, weights_dir="", outputs_dir="o") == []  # tuatara.cpp:315-318
    assert "Please provide a value for weights_dir" in capfd.readouterr().err
    assert pytuatara.image_to_data(img, "w", "") == []                                 # tuatara.cpp:320-323
    assert "Please provide a value for outputs_dir" in capfd.readouterr().err
    assert pytuatara.image_to_data(img, "/nonexistent/weights", "o") == []             # tuatara.cpp:337-340
    assert "error loading" in capfd.readouterr().err


def test_tokenizer_host_matches_oracle(built):
    from oracle import post
    from tuatara_amd.engine import decode_ids
    rng = np.random.default_rng(0)
    for _ in range(200):
        ids = rng.integers(0, 95, 26)
        ids[rng.integers(0, 26)] = 0
        if rng.random() < 0.3:
            ids[rng.integers(0, 26)] = 88
        assert decode_ids(ids) == post.decode_ids(ids)


def test_host_min_area_rect_matches_oracle(built):
    """Engine: float32 rotating calipers.  Oracle: exhaustive double-precision search.  Same rectangle."""
    from oracle import post
    from tuatara_amd.engine import min_area_rect
    rng = np.random.default_rng(1)
    for i in range(60):
        th = rng.uniform(0, np.pi) if i % 2 else 0.0
        pts = rng.normal(size=(40, 2)) * [rng.uniform(5, 40), rng.uniform(2, 8)]
        pts = np.round(pts @ np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]]) + 200)
        a, b = min_area_rect(pts), post.min_area_rect(pts)
        ca, cb = np.sort(post.rect_points(a), 0), np.sort(post.rect_points(b), 0)
        assert abs(a[2] * a[3] - b[2] * b[3]) <= 1e-3 * b[2] * b[3]
        assert np.abs(ca - cb).max() < 2e-2, (a, b)


def test_host_component_rect_matches_oracle_dilation(built):
    """Row-extreme dilation shortcut (engine) vs full-image dilate + findNonZero (oracle)."""
    from scipy import ndimage as ndi
    from oracle import post
    from tuatara_amd.engine import component_rect
    rng = np.random.default_rng(2)
    H, W = 64, 96
    for trial in range(25):
        text = np.zeros((H, W), np.float32)
        cy, cx = rng.integers(2, H - 2), rng.integers(2, W - 2)      # includes blobs touching the borders
        yy, xx = np.mgrid[0:H, 0:W]
        th = rng.uniform(-0.6, 0.6)
        u, v = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th), -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        text = np.exp(-0.5 * ((u / rng.uniform(4, 14)) ** 2 + (v / rng.uniform(1.5, 5)) ** 2)).astype(np.float32)
        link = np.zeros_like(text)
        link[0, 0] = 1.0
        rects, labels, tn = post.get_detected_boxes(text, link)
        if len(rects) != 1:
            continue
        k = labels[cy, cx]
        m = labels == k
        ys, xs = np.nonzero(m)
        x0, x1, y0, y1 = xs.min(), xs.max(), ys.min(), ys.max()
        rows = np.array([[np.nonzero(m[y])[0].min(), np.nonzero(m[y])[0].max()] for y in range(y0, y1 + 1)], np.int32)
        got = component_rect(int(m.sum()), x0, y0, x1, y1, rows, H, W)
        ca, cb = np.sort(post.rect_points(got), 0), np.sort(post.rect_points(rects[0]), 0)
        assert np.abs(ca - cb).max() < 2e-2, (trial, got, rects[0])


def test_host_box_geometry_matches_oracle(built):
    from oracle import post
    from tuatara_amd.engine import box_geometry
    g = np.load(os.path.join(ROOT, "tests", "golden", "g7_funsd.npz"))
    for r in g["det"]:
        adj, xywh, bbox = box_geometry(r, 1.0)
        o = post.adjust_result_coordinates(r[None], 1.0, 1.0)[0]
        assert np.abs(np.sort(post.rect_points(adj), 0) - np.sort(post.rect_points(o), 0)).max() < 1e-2
        assert xywh == post.bounding_rect(o) and bbox == post.tesseract_bbox(o)
