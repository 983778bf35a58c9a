"""ctypes binding + build recipe for oracle/post.c (CPU oracle, TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED — see the header of post.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_post.so")
_SRC = os.path.join(_HERE, "post.c")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-shared", "-fPIC", "-o", _SO, _SRC, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_get_detected_boxes.restype = C.c_int
        _lib.orc_connected_components4.restype = C.c_int
        _lib.orc_crop_resize.restype = C.c_int
        _lib.orc_decode_ids.restype = C.c_int
        _lib.orc_tokenizer_table.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def tokenizer_table():
    buf = C.create_string_buffer(100)
    e, b, p = C.c_int(), C.c_int(), C.c_int()
    n = lib().orc_tokenizer_table(buf, C.byref(e), C.byref(b), C.byref(p))
    return buf.raw[:n].decode("latin1"), e.value, b.value, p.value


def decode_ids(ids) -> str:
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    out = C.create_string_buffer(len(ids) + 1)
    lib().orc_decode_ids(_p(ids, C.c_int64), C.c_int(len(ids)), out)
    return out.value.decode("latin1")


def decode_logits(logits: np.ndarray):
    """[N,L,C] fp32 logits -> (strings, ids[N,L]) exactly as tuatara.cpp:486-505."""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    N, L, Cc = logits.shape
    ids = np.zeros((N, L), np.int64)
    lib().orc_softmax_argmax(_p(logits, C.c_float), C.c_int(N * L), C.c_int(Cc), _p(ids, C.c_int64), None)
    return [decode_ids(ids[i]) for i in range(N)], ids


def resize_linear(img: np.ndarray, dh: int, dw: int) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros((dh, dw, 3), np.uint8)
    lib().orc_resize_linear_u8c3(_p(img, C.c_uint8), C.c_int(img.shape[0]), C.c_int(img.shape[1]), C.c_int(img.shape[1] * 3),
                                 _p(out, C.c_uint8), C.c_int(dh), C.c_int(dw), C.c_int(dw * 3))
    return out


def resize_aspect_ratio_dims(h: int, w: int, square: int = 1024, mag: float = 1.0):
    th, tw, th32, tw32, r = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_float()
    lib().orc_resize_aspect_ratio_dims(C.c_int(h), C.c_int(w), C.c_int(square), C.c_float(mag), C.byref(th), C.byref(tw),
                                       C.byref(th32), C.byref(tw32), C.byref(r))
    return th.value, tw.value, th32.value, tw32.value, r.value


def resize_aspect_ratio(img: np.ndarray, square: int = 1024, mag: float = 1.0):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    th, tw, th32, tw32, r = resize_aspect_ratio_dims(h, w, square, mag)
    out = np.zeros((th32, tw32, 3), np.uint8)
    lib().orc_resize_aspect_ratio(_p(img, C.c_uint8), C.c_int(h), C.c_int(w), C.c_int(w * 3), C.c_int(square), C.c_float(mag),
                                  _p(out, C.c_uint8))
    return out, r


def connected_components4(img: np.ndarray):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    labels = np.zeros((H, W), np.int32)
    stats = np.zeros((H * W // 2 + 2, 5), np.int32)
    n = lib().orc_connected_components4(_p(img, C.c_uint8), C.c_int(H), C.c_int(W), _p(labels, C.c_int32), _p(stats, C.c_int32))
    return n, labels, stats[:n].copy()


def get_detected_boxes(textmap: np.ndarray, linkmap: np.ndarray, text_threshold=0.7, link_threshold=0.4, low_text=0.4,
                       max_rects: int = 8192):
    """Returns (rects[n,5] float32 {cx,cy,w,h,angle}, labels[H,W] int32, textmap_norm[H,W])."""
    t = np.ascontiguousarray(textmap, dtype=np.float32)
    l = np.ascontiguousarray(linkmap, dtype=np.float32)
    H, W = t.shape
    rects = np.zeros((max_rects, 5), np.float32)
    labels = np.zeros((H, W), np.int32)
    tn = np.zeros((H, W), np.float32)
    n = lib().orc_get_detected_boxes(_p(t, C.c_float), _p(l, C.c_float), C.c_int(1), C.c_int(H), C.c_int(W),
                                     C.c_float(text_threshold), C.c_float(link_threshold), C.c_float(low_text),
                                     _p(rects, C.c_float), C.c_int(max_rects), _p(labels, C.c_int32), _p(tn, C.c_float))
    assert n <= max_rects
    return rects[:n].copy(), labels, tn


def adjust_result_coordinates(rects: np.ndarray, ratio_w: float, ratio_h: float) -> np.ndarray:
    rects = np.ascontiguousarray(rects, dtype=np.float32).reshape(-1, 5)
    out = np.zeros_like(rects)
    lib().orc_adjust_result_coordinates(_p(rects, C.c_float), C.c_int(len(rects)), C.c_float(ratio_w), C.c_float(ratio_h),
                                        _p(out, C.c_float))
    return out


def rect_points(rect) -> np.ndarray:
    r = np.ascontiguousarray(rect, dtype=np.float32)
    out = np.zeros(8, np.float32)
    lib().orc_rect_points(_p(r, C.c_float), _p(out, C.c_float))
    return out.reshape(4, 2)


def bounding_rect(rect):
    r = np.ascontiguousarray(rect, dtype=np.float32)
    out = np.zeros(4, np.int32)
    lib().orc_bounding_rect(_p(r, C.c_float), _p(out, C.c_int32))
    return tuple(int(v) for v in out)


def tesseract_bbox(rect):
    r = np.ascontiguousarray(rect, dtype=np.float32)
    out = np.zeros(4, np.float32)
    lib().orc_tesseract_bbox(_p(r, C.c_float), _p(out, C.c_float))
    return [float(v) for v in out]


def min_area_rect(points, exhaustive: bool = False) -> np.ndarray:
    """exhaustive=True: double-precision search over all hull edges (independent cross-check)."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
    out = np.zeros(5, np.float32)
    fn = lib().orc_min_area_rect_exhaustive if exhaustive else lib().orc_min_area_rect
    fn(_p(pts, C.c_double), C.c_int(len(pts)), _p(out, C.c_float))
    return out


def crop_resize(image: np.ndarray, rect, clamp: bool = True):
    """image: the channel-swapped page (after tuatara.cpp:349).  Returns [32,128,3] u8 in
    the order PARSeq sees (after the second swap, :441), or None if the crop is invalid."""
    image = np.ascontiguousarray(image, dtype=np.uint8)
    r = np.ascontiguousarray(rect, dtype=np.float32)
    out = np.zeros((32, 128, 3), np.uint8)
    rc = lib().orc_crop_resize(_p(image, C.c_uint8), C.c_int(image.shape[0]), C.c_int(image.shape[1]), C.c_int(image.shape[1] * 3),
                               _p(r, C.c_float), C.c_int(1 if clamp else 0), _p(out, C.c_uint8))
    return out if rc == 0 else None
