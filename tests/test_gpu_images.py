"""-m gpu: image_to_data over a LIST of host images through every layer of the drop-in surface: the C ABI
(ttr_images_to_data), the ctypes wrapper, the pybind11 module (pytuatara.images_to_data) - in the DEFAULT precision.

The reference takes one image per call and reloads both models each time (/root/reference/tuatara.cpp:336, :428; callers loop:
bindings/run_ocr.py:92).  The list form buckets images by size, runs the buckets as streamed batches with pinned, double-buffered staging on
an upload stream, and returns results in input order.  The bar: every image's result equals its single-image call's, which equals the oracle's."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pages(funsd):
    """13 images of five sizes, interleaved: FUNSD (1000 x 754), three synthetic 1024 x 768 pages, crops of FUNSD at 763 x 607 and 206 x 275, a wide
    664 x 1245 montage (resized by 0.82: the fixed-point resize path)."""
    from tuatara_amd import synth
    f = funsd
    wide = np.full((664, 1245, 3), 255, np.uint8)
    wide[:664, :754] = f[100:764]
    wide[:664, 754:1245] = f[200:864, 100:591]
    sizes = {
        "funsd": [f, np.ascontiguousarray(f[::-1, ::-1])],
        "synth": [synth.synthetic_page(200 + i, 1024, 768, n_words=22) for i in range(3)],
        "resume": [np.ascontiguousarray(f[100:863, 60:667]), np.ascontiguousarray(f[200:963, 100:707]), np.ascontiguousarray(f[0:763, 0:607])],
        "small": [np.ascontiguousarray(f[300:506, 200:475]), np.ascontiguousarray(f[500:706, 100:375]), np.ascontiguousarray(f[40:246, 300:575])],
        "wide": [wide, np.ascontiguousarray(wide[:, ::-1])],
    }
    order = ["small", "funsd", "synth", "wide", "resume", "synth", "small", "resume", "funsd", "wide", "synth", "small", "resume"]
    it = {k: iter(v) for k, v in sizes.items()}
    pages = [next(it[k]) for k in order]
    assert len({p.shape for p in pages}) == 5
    return pages


def _same(a, b):
    return [x["text"] for x in a] == [x["text"] for x in b] and [list(x["bbox"]) for x in a] == [list(x["bbox"]) for x in b]


def test_images_of_five_sizes_equal_their_single_calls_and_the_oracle(eng_x4, oracle_models, funsd):
    from oracle import pipeline
    pages = _pages(funsd)
    singles = [eng_x4.image_to_data(p) for p in pages]
    got = eng_x4.images_to_data(pages)
    assert len(got) == len(pages)
    for i, (g, s) in enumerate(zip(got, singles)):
        assert _same(g, s), f"image {i} {pages[i].shape}"
    assert sum(len(g) for g in got) > 300
    for i in (0, 3, 4, 6):                                        # one image of four of the sizes against the CPU oracle (the fifth, FUNSD, has its own test)
        ref = pipeline.image_to_data(*oracle_models, pages[i])
        assert _same(got[i], ref), f"image {i} {pages[i].shape} vs oracle"
    # small batches: every bucket is cut into pieces, more batches than staging slots
    assert eng_x4.set_tuning(b"images_batch", 2) == 0
    try:
        again = eng_x4.images_to_data(pages)
    finally:
        eng_x4.set_tuning(b"images_batch", 32)
    for g, s in zip(again, singles):
        assert _same(g, s)
    # and the engine is free for the synchronous calls afterwards (nothing left in flight)
    assert _same(eng_x4.image_to_data(pages[1]), singles[1])


def test_images_list_edge_cases(eng_x4, funsd):
    from tuatara_amd.engine import EngineError
    assert eng_x4.images_to_data([]) == []
    one = eng_x4.images_to_data([funsd[:300, :400]])
    assert len(one) == 1 and _same(one[0], eng_x4.image_to_data(np.ascontiguousarray(funsd[:300, :400])))   # (a non-contiguous view: copied by the wrapper)
    blank = np.full((64, 96, 3), 255, np.uint8)
    res = eng_x4.images_to_data([blank, funsd[:256, :256], blank])
    single_blank = eng_x4.image_to_data(blank)                       # (a flat page: the min-max normalisation of tuatara.cpp:120-121 stretches rounding noise - whatever it gives)
    assert _same(res[0], single_blank) and _same(res[2], single_blank) and len(res[1]) > 0
    with pytest.raises(EngineError):
        eng_x4.images_to_data([np.zeros((10, 10), np.uint8)])
    # a failure in the middle of a list leaves nothing in flight: the next call works
    assert len(eng_x4.images_to_data([funsd[:256, :256]])[0]) > 0


def test_images_one_unreadable_entry_fails_alone(eng_x4, funsd):
    """a NULL / empty / short-stride entry in the middle of a list is the reference's "Error reading image from file" (tuatara.cpp:344-347) for
    THAT image - a loop over image_to_data goes on with the next one.  ttr_images_to_data returns the count of failed images, leaves their results empty and
    delivers every other result; ttr_last_error lists the indices."""
    import ctypes as C
    a, b = np.ascontiguousarray(funsd[:256, :256]), np.ascontiguousarray(funsd[300:556, 100:356])
    n = 4
    ptrs = (C.c_void_p * n)(a.ctypes.data, None, b.ctypes.data, b.ctypes.data)
    hs = (C.c_int32 * n)(256, 256, 256, 256)
    ws = (C.c_int32 * n)(256, 256, 256, 256)
    st = (C.c_int32 * n)(768, 768, 768, 100)          # entry 3: a stride shorter than a row
    out = (C.c_void_p * n)()
    rc = eng_x4.lib.ttr_images_to_data(eng_x4.h, ptrs, hs, ws, st, n, out)
    msg = eng_x4.lib.ttr_last_error().decode()
    assert rc == 2, (rc, msg)
    assert "2 of 4 images failed (indices 1 3)" in msg and "Error reading image" in msg, msg
    got = eng_x4._take_many(out, n)
    assert _same(got[0], eng_x4.image_to_data(a)) and _same(got[2], eng_x4.image_to_data(b)) and len(got[0]) > 0
    assert got[1] == [] and got[3] == []


def test_images_row_strides_through_the_c_abi(eng_x4, funsd):
    """ttr_images_to_data with row strides: images that are windows of a larger buffer, not copied by the caller."""
    import ctypes as C
    big = np.ascontiguousarray(funsd)
    views = [(0, 0, 400, 300), (200, 100, 400, 300), (500, 300, 256, 384)]
    n = len(views)
    ptrs = (C.c_void_p * n)(*[big.ctypes.data + (y * big.shape[1] + x) * 3 for y, x, h, w in views])
    hs = (C.c_int32 * n)(*[v[2] for v in views])
    ws = (C.c_int32 * n)(*[v[3] for v in views])
    st = (C.c_int32 * n)(*[big.shape[1] * 3] * n)
    out = (C.c_void_p * n)()
    assert eng_x4.lib.ttr_images_to_data(eng_x4.h, ptrs, hs, ws, st, n, out) == 0, eng_x4.lib.ttr_last_error()
    got = eng_x4._take_many(out, n)
    for g, (y, x, h, w) in zip(got, views):
        assert _same(g, eng_x4.image_to_data(np.ascontiguousarray(big[y:y + h, x:x + w])))


def test_pytuatara_images_to_data(weights, funsd, monkeypatch):
    """The drop-in module's list form, default precision: same kwarg style and return structure as image_to_data, one list per image."""
    from tuatara_amd import build
    build.build_pytuatara()
    sys.path.insert(0, os.path.join(ROOT, "build", "bindings"))
    import pytuatara
    monkeypatch.delenv("TUATARA_PRECISION", raising=False)
    pages = _pages(funsd)[:7]
    res = pytuatara.images_to_data(images=pages, weights_dir=weights["dir"], outputs_dir="../outputs")
    assert isinstance(res, list) and len(res) == len(pages)
    for p, r in zip(pages, res):
        one = pytuatara.image_to_data(p, weights["dir"], "../outputs")
        assert isinstance(r, list) and _same(r, one)
        if r:
            assert set(r[0].keys()) == {"text", "bbox"}
    with pytest.raises(RuntimeError, match="3 dimensions"):
        pytuatara.images_to_data([np.zeros((4, 4), np.uint8)], weights["dir"], "../outputs")
    assert pytuatara.images_to_data([], weights["dir"], "../outputs") == []
    assert pytuatara.images_to_data(pages[:1], "", "x") == []                 # the reference's error convention: message on stderr, empty result
