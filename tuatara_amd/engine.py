"""ctypes binding of the C ABI in include/tuatara_hip.h (libtuatara_hip.so).

Thin by design: numpy arrays in, numpy arrays / python lists out; every compute call
runs the HIP engine.  There is no CPU fallback — loading fails loudly when the shared
library is missing and ``Engine(...)`` raises when no GPU is present.
"""
from __future__ import annotations

import collections.abc
import ctypes as C
import os
import sys
from typing import List, Optional, Sequence

import numpy as np

_LIBPATH = os.environ.get("TUATARA_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtuatara_hip.so")   # (TUATARA_LIB: another build of the same ABI, for same-box A/B timing)
_lib = None

PREC_BF16, PREC_F32, PREC_F16X4 = 0, 1, 2


class Config(C.Structure):
    _fields_ = [("precision", C.c_int), ("device", C.c_int), ("canvas_size", C.c_int), ("mag_ratio", C.c_float),
                ("text_threshold", C.c_float), ("link_threshold", C.c_float), ("low_text", C.c_float), ("min_area", C.c_int),
                ("strict_crops", C.c_int), ("max_components", C.c_int), ("verbose", C.c_int)]


# every symbol include/tuatara_hip.h declares: (name, restype, argtypes)
_VP, _I, _F = C.c_void_p, C.c_int, C.c_float
_PU8, _PF, _PI = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32)
SYMBOLS = [
    ("ttr_config_default", None, [C.POINTER(Config)]),
    ("ttr_create", _VP, [C.c_char_p, C.POINTER(Config)]),
    ("ttr_destroy", None, [_VP]),
    ("ttr_last_error", C.c_char_p, []),
    ("ttr_version", C.c_char_p, []),
    ("ttr_image_to_data", _I, [_VP, _PU8, _I, _I, _I, C.POINTER(_VP)]),
    ("ttr_pages_to_data_dev", _I, [_VP, _VP, _I, _I, _I, C.POINTER(_VP)]),
    ("ttr_stream_push", _I, [_VP, _VP, _I, _I, _I, C.POINTER(_VP), C.POINTER(C.c_int)]),
    ("ttr_stream_flush", _I, [_VP, C.POINTER(_VP), C.POINTER(C.c_int)]),
    ("ttr_images_to_data", _I, [_VP, C.POINTER(_VP), _PI, _PI, _PI, _I, C.POINTER(_VP)]),
    ("ttr_result_count", _I, [_VP]),
    ("ttr_result_text", C.c_char_p, [_VP, _I]),
    ("ttr_result_bbox", _PF, [_VP, _I]),
    ("ttr_result_ids", _PI, [_VP, _I]),
    ("ttr_result_free", None, [_VP]),
    ("ttr_result_bboxes", _PF, [_VP]),
    ("ttr_result_ids_all", _PI, [_VP]),
    ("ttr_result_texts", _I, [_VP, C.c_char_p, C.c_size_t]),
    ("ttr_results_gather", _I, [C.POINTER(_VP), _I, _PI, _PF, _PI, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("ttr_craft_heatmap", _I, [_VP, _PU8, _I, _I, _PF]),
    ("ttr_ccl_boxes", _I, [_VP, _PF, _I, _I, _PF, _I, _PI]),
    ("ttr_resize_canvas", _I, [_VP, _PU8, _I, _I, _I, _PU8, C.c_size_t, _PI, _PI, _PF]),
    ("ttr_pack_crops", _I, [_VP, _PU8, _I, _I, _I, _PF, _I, _F, _PU8, _PF]),
    ("ttr_parseq_logits", _I, [_VP, _PU8, _I, _PF, _PF, _PI]),
    ("ttr_decode_ids", _I, [_PI, _I, C.c_char_p]),
    ("ttr_engine_set_tuning", _I, [_VP, C.c_char_p, _I]),
    ("ttr_dbg_conv", _I, [_VP, _PF, _I, _PF, _I, _I, _I, _I, _I, _I, _I, _I, _PF, _PF, _I, _I, _PF]),
    ("ttr_dbg_min_area_rect", _I, [_PF, _I, _PF]),
    ("ttr_dbg_tcp_share", _I, [_I, _I, C.c_char_p, _I, _VP, C.c_size_t]),
    ("ttr_dbg_component_rect", _I, [_I, _I, _I, _I, _I, _PI, _I, _I, _PF]),
    ("ttr_dbg_box_geometry", _I, [_PF, _F, _PF, _PI, _PF]),
    ("ttr_dev_alloc", _VP, [C.c_size_t]),
    ("ttr_dev_free", None, [_VP]),
    ("ttr_dev_upload", _I, [_VP, _VP, C.c_size_t]),
    ("ttr_dev_download", _I, [_VP, _VP, C.c_size_t]),
    ("ttr_dev_sync", _I, [_VP]),
    ("ttr_last_stage_ms", _I, [_VP, _PF]),
    ("ttr_set_profiling", _I, [_VP, _I]),
    ("ttr_dbg_conv_pool", _I, [_VP, _PF, _I, _I, _I, _I, _I, _PF, _PF, _I, _I, _I, _PF, _PF]),
    ("ttr_dbg_split_gemm", _I, [_VP, _PF, _I, _I, _PF, _PF, _I, _I, _I, _I, _PF, _I, _PF]),
    ("ttr_set_gemm_config", None, [_I]),
    ("ttr_set_decoder_mode", None, [_I]),
    ("ttr_set_tuning", _I, [C.c_char_p, _I]),
    ("ttr_last_host_us", None, [_VP, _PF]),
    ("ttr_dbg_attn_enc", _I, [_VP, _PF, _I, _PF]),
    ("ttr_dbg_cross_attn", _I, [_VP, _PF, _PF, _I, _I, _PF]),
    ("ttr_dbg_qkv_attn", _I, [_VP, _PF, _I, _PF, _PF, _PF]),
    ("ttr_dbg_mlp", _I, [_VP, _PF, _I, _PF, _PF, C.c_float, _PF, _PF, _PF, _PF, _PF, _PF, _PF, _PF, _PF, _PF, _PF]),
    ("ttr_dbg_dec_stamps", _I, [C.POINTER(C.c_ulonglong)]),
    ("ttr_dbg_dec_stamps_ext", _I, [C.POINTER(C.c_ulonglong), _I]),
    ("ttr_bench_conv", _I, [_VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _PF]),
    ("ttr_get_profile", _I, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    ("ttr_get_profile_kinds", _I, [_VP, C.c_char_p, C.c_size_t]),
    ("ttr_comm_unique_id", _I, [_VP]),
    ("ttr_comm_create", _VP, [_VP, _I, _I, _VP]),
    ("ttr_comm_create_tcp", _VP, [_VP, _I, _I, C.c_char_p, _I]),
    ("ttr_comm_create_socket", _VP, [_VP, _I, _I, C.c_char_p, _I]),
    ("ttr_comm_transport", C.c_char_p, [_VP]),
    ("ttr_comm_describe", _I, [_VP, C.c_char_p, C.c_size_t]),
    ("ttr_comm_destroy", None, [_VP]),
    ("ttr_comm_rank", _I, [_VP]),
    ("ttr_comm_world", _I, [_VP]),
    ("ttr_engine_attach_comm", _I, [_VP, _VP]),
    ("ttr_last_gathered", _I, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int), _PI, C.c_size_t, _PI, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("ttr_comm_allgather_host", _I, [_VP, _VP, C.c_size_t, _VP]),
    ("ttr_gather_layout", _I, [_PI, _I, _I, C.POINTER(C.c_int), _PI, C.POINTER(C.c_int64)]),
    ("ttr_pages_to_data_dev_sharded", _I, [_VP, _VP, _I, _I, _I, C.POINTER(_VP)]),
]


def lib_path() -> str:
    return _LIBPATH


def load():
    """Load libtuatara_hip.so (no GPU needed for loading)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise RuntimeError(f"{_LIBPATH} is missing: run `python -m tuatara_amd.build` (or __graft_entry__.build()) first")
        # PyTorch's ROCm wheel carries its own HIP runtime.  If torch is imported AFTER this library (which links the system
        # runtime) has been loaded, the process ends up with two runtimes and the one that initialises second sees no device
        # ("no HIP device available").  Imported first, torch's copy is the one both use.  A Python process that loads the
        # engine may import torch later (the oracle, torch.distributed), so: torch first, where it is installed
        # (TUATARA_PRELOAD_TORCH=0 turns this off; the C++ callers - pytuatara, ocr_cli - never see torch).
        if "torch" not in sys.modules and os.environ.get("TUATARA_PRELOAD_TORCH", "1") != "0":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(_LIBPATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _u8(a):
    return a.ctypes.data_as(_PU8)


def _f(a):
    return a.ctypes.data_as(_PF)


def _i(a):
    return a.ctypes.data_as(_PI)


class EngineError(RuntimeError):
    pass


def decode_ids(ids: Sequence[int]) -> str:
    a = np.ascontiguousarray(ids, dtype=np.int32)
    buf = C.create_string_buffer(len(a) + 2)
    load().ttr_decode_ids(_i(a), len(a), buf)
    return buf.value.decode("latin1")


def min_area_rect(points) -> np.ndarray:
    """Engine host geometry (no GPU): cv::minAreaRect stand-in."""
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 2)
    out = np.zeros(5, np.float32)
    load().ttr_dbg_min_area_rect(_f(pts), len(pts), _f(out))
    return out


def component_rect(area: int, x0: int, y0: int, x1: int, y1: int, rows: np.ndarray, H: int, W: int):
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    out = np.zeros(5, np.float32)
    ok = load().ttr_dbg_component_rect(area, x0, y0, x1, y1, _i(rows), H, W, _f(out))
    return out if ok == 1 else None


def box_geometry(rect5, ratio: float):
    r = np.ascontiguousarray(rect5, dtype=np.float32)
    adj, xywh, bbox = np.zeros(5, np.float32), np.zeros(4, np.int32), np.zeros(4, np.float32)
    load().ttr_dbg_box_geometry(_f(r), C.c_float(ratio), _f(adj), _i(xywh), _f(bbox))
    return adj, tuple(int(v) for v in xywh), [float(v) for v in bbox]


class PageResult(collections.abc.Sequence):
    """One page's words as the list of {"text", "bbox", "ids"} dicts pytuatara.image_to_data returns, materialised on access:
    the batch hand-over keeps the arrays the C ABI filled (`texts`, `bbox` f32 [n,4], `ids` i32 [n,26]) and builds dicts only
    for the items a caller touches."""
    __slots__ = ("texts", "bbox", "ids")

    def __init__(self, texts, bbox, ids):
        self.texts, self.bbox, self.ids = texts, bbox, ids

    def __len__(self):
        return len(self.texts)

    def __getitem__(self, j):
        if isinstance(j, slice):
            return [self[i] for i in range(*j.indices(len(self)))]
        if j < 0:
            j += len(self)
        if not 0 <= j < len(self):
            raise IndexError(j)
        return {"text": self.texts[j], "bbox": self.bbox[j].tolist(), "ids": self.ids[j].tolist()}

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return repr(list(self))


class DeviceBuffer:
    def __init__(self, nbytes: int):
        self.ptr = load().ttr_dev_alloc(nbytes)
        if not self.ptr:
            raise EngineError("device allocation failed")
        self.nbytes = nbytes

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        if load().ttr_dev_upload(self.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes) != 0:
            raise EngineError("upload failed")

    def free(self):
        if self.ptr:
            load().ttr_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    def __init__(self, weights_dir: str, precision: str = "f16x4", device: int = 0, strict_crops: bool = False, **overrides):
        self.lib = load()
        # A process that also uses torch's GPU runtime (bench.py, tuatara_amd/dist.py over RCCL) must let torch initialise FIRST:
        # the torch wheel bundles its own ROCm 7.0 HIP / HSA libraries, and they do not come up once the system ROCm 7.2 runtime the
        # engine links has claimed the device ("No HIP GPUs are available"); the other order works.  So if torch is already
        # imported, bring its runtime up before the engine's.
        _t = sys.modules.get("torch")
        if _t is not None and hasattr(_t, "cuda"):
            try:
                if _t.cuda.is_available():
                    _t.cuda.init()
            except Exception:
                pass
        cfg = Config()
        self.lib.ttr_config_default(C.byref(cfg))
        cfg.precision = (PREC_F32 if precision in ("f32", "fp32", PREC_F32) else
                         PREC_F16X4 if precision in ("f16x4", "split", PREC_F16X4) else PREC_BF16)
        cfg.device = device
        cfg.strict_crops = int(strict_crops)
        tuning = {k: overrides.pop(k) for k in list(overrides) if not hasattr(cfg, k)}     # not a config field: a tuning key (below)
        for k, v in overrides.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.h = self.lib.ttr_create(weights_dir.encode(), C.byref(cfg))
        if not self.h:
            raise EngineError(self.lib.ttr_last_error().decode())
        for k, v in tuning.items():
            if self.set_tuning(k, int(v)) != 0:
                raise EngineError(f"unknown engine option {k!r}")

    def set_tuning(self, key, value: int) -> int:
        """Per-engine kernel-selection knob (ttr_engine_set_tuning); keys it does not know go to the process-wide diagnostics setter."""
        return self.lib.ttr_engine_set_tuning(self.h, key if isinstance(key, bytes) else key.encode(), int(value))

    def close(self):
        if getattr(self, "h", None):
            self.lib.ttr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(self.lib.ttr_last_error().decode())

    def _take(self, r) -> List[dict]:
        """ttr_result -> the reference's list of {"text", "bbox"} dicts (+ "ids"), through the bulk getters."""
        n = self.lib.ttr_result_count(r)
        out = []
        if n:
            bb = np.ctypeslib.as_array(self.lib.ttr_result_bboxes(r), (n, 4)).tolist()
            ids = np.ctypeslib.as_array(self.lib.ttr_result_ids_all(r), (n, 26)).tolist()
            need = self.lib.ttr_result_texts(r, None, 0)
            buf = C.create_string_buffer(need)
            self.lib.ttr_result_texts(r, buf, need)
            texts = buf.raw[:need].decode("latin1").split("\n")
            out = [{"text": texts[i], "bbox": bb[i], "ids": ids[i]} for i in range(n)]
        self.lib.ttr_result_free(r)
        return out

    def _take_many(self, arr, n: int) -> List[List[dict]]:
        """A batch of ttr_results -> list (per page) of lists of {"text", "bbox", "ids"}: one gather call for the whole batch."""
        counts = np.zeros(n, np.int32)
        need = C.c_size_t()
        total = self.lib.ttr_results_gather(arr, n, _i(counts), None, None, None, 0, C.byref(need))
        bb = np.zeros((max(total, 1), 4), np.float32)
        ids = np.zeros((max(total, 1), 26), np.int32)
        buf = C.create_string_buffer(max(need.value, 1))
        self.lib.ttr_results_gather(arr, n, None, _f(bb), _i(ids), buf, need.value, None)
        texts = buf.raw[:need.value].decode("latin1").split("\n")
        out, k = [], 0
        for i in range(n):
            c = int(counts[i])
            out.append(PageResult(texts[k:k + c], bb[k:k + c], ids[k:k + c]))
            k += c
            self.lib.ttr_result_free(arr[i])
        return out

    # ---- hot path
    def image_to_data(self, image: np.ndarray) -> List[dict]:
        if image.ndim != 3:
            raise RuntimeError("Input array should have 3 dimensions")          # bindings/python.cpp:15-17 of the reference
        if image.shape[2] != 3:
            raise RuntimeError("Input array should have 3 channels")            # the C ABI reads rows of 3 * w bytes
        image = np.ascontiguousarray(image, dtype=np.uint8)
        r = C.c_void_p()
        self._check(self.lib.ttr_image_to_data(self.h, _u8(image), image.shape[0], image.shape[1], image.shape[1] * 3, C.byref(r)))
        return self._take(r)

    def images_to_data(self, images, keep: bool = True):
        """image_to_data over a list of host images [H, W, 3] u8 of any sizes (ttr_images_to_data): one result list per image, input order."""
        arrs = []
        for im in images:
            a = np.asarray(im)
            if a.ndim != 3 or a.shape[2] != 3:
                raise EngineError("Input array should have 3 dimensions")
            arrs.append(np.ascontiguousarray(a, dtype=np.uint8))
        n = len(arrs)
        if n == 0:
            return []
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        hs = (C.c_int32 * n)(*[a.shape[0] for a in arrs])
        ws = (C.c_int32 * n)(*[a.shape[1] for a in arrs])
        out = (C.c_void_p * n)()
        rc = self.lib.ttr_images_to_data(self.h, ptrs, hs, ws, None, n, out)
        self.last_images_error = None
        if rc < 0:
            self._check(rc)
        if rc > 0:                              # rc images failed: their results are empty, the others delivered (include/tuatara_hip.h)
            self.last_images_error = self.lib.ttr_last_error().decode()
        if keep:
            return self._take_many(out, n)
        counts = []
        for i in range(n):
            counts.append(self.lib.ttr_result_count(out[i]))
            self.lib.ttr_result_free(out[i])
        return counts

    def pages_to_data_dev(self, d_pages, n: int, h: int, w: int, keep: bool = True):
        """d_pages: DeviceBuffer or raw device pointer holding [n][h][w][3] u8."""
        ptr = d_pages.ptr if isinstance(d_pages, DeviceBuffer) else d_pages
        arr = (C.c_void_p * n)()
        self._check(self.lib.ttr_pages_to_data_dev(self.h, ptr, n, h, w, arr))
        if keep:
            return self._take_many(arr, n)
        counts = []
        for i in range(n):
            counts.append(self.lib.ttr_result_count(arr[i]))
            self.lib.ttr_result_free(arr[i])
        return counts

    def _stream_take(self, arr, n_prev: int, keep: bool):
        if keep:
            return self._take_many(arr, n_prev)
        counts = []
        for i in range(n_prev):
            counts.append(self.lib.ttr_result_count(arr[i]))
            self.lib.ttr_result_free(arr[i])
        return counts

    def stream_push(self, d_pages, n: int, h: int, w: int, keep: bool = True, max_batch: int = 0):
        """Streamed batches (ttr_stream_push): enqueue batch k+1, get batch k's results (an empty list on the first push).  The
        pages of a batch must stay alive until its results have come back."""
        ptr = d_pages.ptr if isinstance(d_pages, DeviceBuffer) else d_pages
        self._max_pushed = max(getattr(self, "_max_pushed", 1), n, max_batch)   # a returned batch is never larger than the largest pushed
        arr = (C.c_void_p * self._max_pushed)()
        n_prev = C.c_int(0)
        self._check(self.lib.ttr_stream_push(self.h, ptr, n, h, w, arr, C.byref(n_prev)))
        return self._stream_take(arr, n_prev.value, keep)

    def stream_flush(self, keep: bool = True):
        arr = (C.c_void_p * getattr(self, "_max_pushed", 1))()
        n_prev = C.c_int(0)
        self._check(self.lib.ttr_stream_flush(self.h, arr, C.byref(n_prev)))
        return self._stream_take(arr, n_prev.value, keep)

    def last_stage_ms(self):
        ms = (C.c_float * 4)()
        self.lib.ttr_last_stage_ms(self.h, ms)
        return dict(craft=ms[0], post=ms[1], pack=ms[2], parseq=ms[3])

    def dbg_attn_enc(self, qkv):
        """Encoder self-attention on qkv [N,128,1152] -> [N,128,384] (values rounded to the engine's type on the way in / out)."""
        qkv = np.ascontiguousarray(qkv, np.float32)
        N = qkv.shape[0]
        out = np.zeros((N, 128, 384), np.float32)
        self._check(self.lib.ttr_dbg_attn_enc(self.h, _f(qkv), N, _f(out)))
        return out

    def dbg_cross_attn(self, q, kvmem):
        """The decoder's cross-attention kernels on their own: q [N, R, 384], kvmem [N, 128, 768] (K | V) -> [N, R, 384] (split / fp32 engines)."""
        q = np.ascontiguousarray(q, np.float32); kvmem = np.ascontiguousarray(kvmem, np.float32)
        N, R = q.shape[0], q.shape[1]
        out = np.zeros((N, R, 384), np.float32)
        self._check(self.lib.ttr_dbg_cross_attn(self.h, _f(q), _f(kvmem), N, R, _f(out)))
        return out

    def dbg_qkv_attn(self, x, w, b):
        """qkv_attn.hip: x [N,128,384] (LayerNorm output), w [1152,384], b [1152] -> attention output [N,128,384]."""
        x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32); b = np.ascontiguousarray(b, np.float32)
        out = np.zeros((x.shape[0], 128, 384), np.float32)
        self._check(self.lib.ttr_dbg_qkv_attn(self.h, _f(x), x.shape[0], _f(w), _f(b), _f(out)))
        return out

    def dbg_mlp(self, x, ln_g, ln_b, w1, b1, w2, b2, nln_g=None, nln_b=None, eps=1e-6, att=None, wp=None, bp=None):
        """mlp_fused.hip on f32 rows x [M,384] -> (x_out f32 [M,384], LayerNorm_next(x_out) as f32 or None)."""
        f = lambda a: np.ascontiguousarray(a, np.float32)
        x, ln_g, ln_b, w1, b1, w2, b2 = map(f, (x, ln_g, ln_b, w1, b1, w2, b2))
        M = x.shape[0]
        out = np.zeros((M, 384), np.float32)
        nout = np.zeros((M, 384), np.float32) if nln_g is not None else None
        ng, nb = (f(nln_g), f(nln_b)) if nln_g is not None else (None, None)
        self._check(self.lib.ttr_dbg_mlp(self.h, _f(x), M, _f(ln_g), _f(ln_b), eps, _f(w1), _f(b1), _f(w2), _f(b2),
                                         _f(ng) if ng is not None else None, _f(nb) if nb is not None else None, _f(out),
                                         _f(nout) if nout is not None else None,
                                         _f(f(att)) if att is not None else None, _f(f(wp)) if att is not None else None, _f(f(bp)) if att is not None else None))
        return out, nout

    def last_host_us(self):
        us = (C.c_float * 8)()
        self.lib.ttr_last_host_us(self.h, us)
        return [round(float(x), 1) for x in us]

    def set_profiling(self, on):
        """0 / False off, 1 / True CRAFT conv launches only, 2 every conv / GEMM launch."""
        self.lib.ttr_set_profiling(self.h, int(on))

    def get_profile(self):
        ms, fl, n = (C.c_double * 3)(), (C.c_double * 3)(), (C.c_longlong * 3)()
        self.lib.ttr_get_profile(self.h, ms, fl, n)
        return {"craft": dict(ms=ms[0], flops=fl[0], launches=n[0]), "parseq": dict(ms=ms[1], flops=fl[1], launches=n[1]),
                "parseq_ar": dict(ms=ms[2], flops=fl[2], launches=n[2])}

    def get_profile_kinds(self):
        """The timed launches by kernel kind: list of dicts {kind, stage, launches, ms, alg_flops, exec_flops} (ttr_get_profile_kinds)."""
        import json
        buf = C.create_string_buffer(1 << 16)
        n = self.lib.ttr_get_profile_kinds(self.h, buf, len(buf))
        if n >= len(buf):                                   # the call returns the full length: come back with room for it
            buf = C.create_string_buffer(n + 1)
            n = self.lib.ttr_get_profile_kinds(self.h, buf, len(buf))
        if n < 0:
            raise EngineError(self.lib.ttr_last_error().decode())
        return json.loads(buf.value.decode())

    # ---- stages
    def craft_heatmap(self, canvas: np.ndarray) -> np.ndarray:
        canvas = np.ascontiguousarray(canvas, dtype=np.uint8)
        H, W = canvas.shape[:2]
        heat = np.zeros((H // 2, W // 2, 2), np.float32)
        self._check(self.lib.ttr_craft_heatmap(self.h, _u8(canvas), H, W, _f(heat)))
        return heat

    def ccl_boxes(self, heat: np.ndarray, max_rects: int = 8192) -> np.ndarray:
        heat = np.ascontiguousarray(heat, dtype=np.float32)
        H2, W2 = heat.shape[:2]
        rects = np.zeros((max_rects, 5), np.float32)
        n = C.c_int32()
        self._check(self.lib.ttr_ccl_boxes(self.h, _f(heat), H2, W2, _f(rects), max_rects, C.byref(n)))
        return rects[: n.value].copy()

    def resize_canvas(self, image: np.ndarray):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        cap = 1056 * 1056 * 3 * 4
        buf = np.zeros(cap, np.uint8)
        H, W, ratio = C.c_int32(), C.c_int32(), C.c_float()
        self._check(self.lib.ttr_resize_canvas(self.h, _u8(image), image.shape[0], image.shape[1], image.shape[1] * 3, _u8(buf), cap,
                                               C.byref(H), C.byref(W), C.byref(ratio)))
        return buf[: H.value * W.value * 3].reshape(H.value, W.value, 3).copy(), ratio.value

    def pack_crops(self, image: np.ndarray, rects: np.ndarray, ratio: float):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        rects = np.ascontiguousarray(rects, dtype=np.float32).reshape(-1, 5)
        n = len(rects)
        crops = np.zeros((n, 32, 128, 3), np.uint8)
        boxes = np.zeros((n, 5), np.float32)
        self._check(self.lib.ttr_pack_crops(self.h, _u8(image), image.shape[0], image.shape[1], image.shape[1] * 3, _f(rects), n,
                                            C.c_float(ratio), _u8(crops), _f(boxes)))
        return crops, boxes

    def parseq_logits(self, crops: np.ndarray, want_ar: bool = False):
        crops = np.ascontiguousarray(crops, dtype=np.uint8)
        n = len(crops)
        logits = np.zeros((n, 26, 95), np.float32)
        ar = np.zeros((n, 26, 95), np.float32) if want_ar else None
        ids = np.zeros((n, 26), np.int32)
        self._check(self.lib.ttr_parseq_logits(self.h, _u8(crops), n, _f(logits), _f(ar) if want_ar else None, _i(ids)))
        return (logits, ar, ids) if want_ar else (logits, ids)

    def dbg_conv_pool(self, x0: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray], ks: int, act: int = 0, pool_relu: bool = False,
                      want_full: bool = True):
        """bf16 engines: conv + fused 2x2 max-pool -> (full f32 [B,H,W,Cout] or None, pooled f32 [B,H/2,W/2,Cout])."""
        x0 = np.ascontiguousarray(x0, dtype=np.float32)
        B, H, W_, C0 = x0.shape
        w = np.ascontiguousarray(w, dtype=np.float32)
        Cout = w.shape[0]
        full = np.zeros((B, H, W_, Cout), np.float32) if want_full else None
        pool = np.zeros((B, H // 2, W_ // 2, Cout), np.float32)
        b = np.ascontiguousarray(bias, dtype=np.float32) if bias is not None else None
        self._check(self.lib.ttr_dbg_conv_pool(self.h, _f(x0), C0, B, H, W_, ks, _f(w), _f(b) if b is not None else None, Cout, act,
                                               int(pool_relu), _f(full) if want_full else None, _f(pool)))
        return full, pool

    def dbg_conv(self, x0: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray], ks: int, dil: int = 1, act: int = 0,
                 x1: Optional[np.ndarray] = None, relu0: bool = False, relu1: bool = False) -> np.ndarray:
        """x0 f32 NHWC [B,H,W,C0], w f32 [Cout,ks,ks,C0+C1] -> f32 NHWC [B,H,W,Cout]."""
        x0 = np.ascontiguousarray(x0, dtype=np.float32)
        B, H, W_, C0 = x0.shape
        C1 = 0
        if x1 is not None:
            x1 = np.ascontiguousarray(x1, dtype=np.float32)
            C1 = x1.shape[-1]
        w = np.ascontiguousarray(w, dtype=np.float32)
        Cout = w.shape[0]
        out = np.zeros((B, H, W_, Cout), np.float32)
        b = np.ascontiguousarray(bias, dtype=np.float32) if bias is not None else None
        self._check(self.lib.ttr_dbg_conv(self.h, _f(x0), C0, _f(x1) if x1 is not None else None, C1, int(relu0), int(relu1), B, H, W_, ks, dil,
                                          _f(w), _f(b) if b is not None else None, Cout, act, _f(out)))
        return out

    def dbg_split_gemm(self, x: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray] = None, np_products: int = 3, act: int = 0, out_planes: int = 0,
                       resid: Optional[np.ndarray] = None, cfg: int = 0) -> np.ndarray:
        """f16x4 engines: one split-operand linear, x f32 [M,K], w f32 [N,K] -> act(x w^T + bias (+ resid)) f32 [M,N] (tests)."""
        x = np.ascontiguousarray(x, dtype=np.float32); w = np.ascontiguousarray(w, dtype=np.float32)
        M, K = x.shape
        N = w.shape[0]
        out = np.zeros((M, N), np.float32)
        b = np.ascontiguousarray(bias, dtype=np.float32) if bias is not None else None
        r = np.ascontiguousarray(resid, dtype=np.float32) if resid is not None else None
        self._check(self.lib.ttr_dbg_split_gemm(self.h, _f(x), M, K, _f(w), _f(b) if b is not None else None, N, np_products, act, out_planes,
                                                _f(r) if r is not None else None, cfg, _f(out)))
        return out


def gather_layout(counts: np.ndarray):
    """The framing of a gathered batch (ttr_gather_layout, host logic only): counts int32 [world, pages] -> (cap, total[world],
    first[world * pages + 1])."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    world, pages = counts.shape
    cap = C.c_int()
    total = np.zeros(world, np.int32)
    first = np.zeros(world * pages + 1, np.int64)
    if load().ttr_gather_layout(_i(counts), world, pages, C.byref(cap), _i(total), first.ctypes.data_as(C.POINTER(C.c_int64))) != 0:
        raise EngineError(load().ttr_last_error().decode())
    return cap.value, total, first


class Comm:
    """RCCL communicator pair of an engine (include/tuatara_hip.h, "multi-GPU"): one process per GPU.  `Comm(engine, rank, world,
    addr, port)`: rank 0 listens on addr:port and hands the NCCL ids to the others; `attach()` makes every batch of the engine
    all-gather its token ids on the engine's stream."""

    def __init__(self, engine: "Engine", rank: int, world: int, addr: str = "127.0.0.1", port: int = 29617, unique_id: Optional[bytes] = None,
                 transport: str = "rccl"):
        self.eng, self.lib = engine, engine.lib
        if transport == "socket":        # TCP through rank 0: ranks that share one GPU (tests), or where RCCL cannot initialise
            self.h = self.lib.ttr_comm_create_socket(engine.h, rank, world, addr.encode(), port)
        elif unique_id is not None:
            buf = C.create_string_buffer(unique_id, 256)
            self.h = self.lib.ttr_comm_create(engine.h, rank, world, buf)
        else:
            self.h = self.lib.ttr_comm_create_tcp(engine.h, rank, world, addr.encode(), port)
        if not self.h:
            raise EngineError(self.lib.ttr_last_error().decode())
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(256)
        if load().ttr_comm_unique_id(buf) != 0:
            raise EngineError(load().ttr_last_error().decode())
        return buf.raw

    def attach(self, on: bool = True):
        if self.lib.ttr_engine_attach_comm(self.eng.h, self.h if on else None) != 0:
            raise EngineError(self.lib.ttr_last_error().decode())

    def allgather_host(self, mine: np.ndarray) -> np.ndarray:
        """Small host array of every rank, stacked by rank (collective; with an empty array: a barrier)."""
        mine = np.ascontiguousarray(mine)
        out = np.zeros((self.world,) + mine.shape, mine.dtype)
        if self.lib.ttr_comm_allgather_host(self.h, mine.ctypes.data_as(C.c_void_p), mine.nbytes, out.ctypes.data_as(C.c_void_p)) != 0:
            raise EngineError(self.lib.ttr_last_error().decode())
        return out

    def barrier(self):
        self.allgather_host(np.zeros(1, np.int32))

    def describe(self) -> dict:
        """this rank's end of the communicator (ttr_comm_describe): rank, world, transport, RCCL version, HIP device, PCI bus id"""
        import json
        buf = C.create_string_buffer(512)
        if self.lib.ttr_comm_describe(self.h, buf, len(buf)) < 0:
            raise EngineError(self.lib.ttr_last_error().decode())
        return json.loads(buf.value.decode())

    def describe_all(self) -> list:
        """every rank's describe(), by rank (collective)"""
        import json
        mine = np.zeros(512, np.uint8)
        raw = json.dumps(self.describe()).encode()[:511]
        mine[:len(raw)] = np.frombuffer(raw, np.uint8)
        return [json.loads(bytes(r).split(b"\0", 1)[0].decode()) for r in self.allgather_host(mine)]

    def last_gathered(self):
        """(counts int32 [world, pages], ids int32 [rows, 26]) of the batch whose results the engine returned last."""
        world, pages, need = C.c_int(), C.c_int(), C.c_size_t()
        self.lib.ttr_last_gathered(self.eng.h, C.byref(world), C.byref(pages), None, 0, None, 0, C.byref(need))
        counts = np.zeros((max(world.value, 0), max(pages.value, 0)), np.int32)
        ids = np.zeros((need.value // 26, 26), np.int32)
        self.lib.ttr_last_gathered(self.eng.h, None, None, _i(counts), counts.size, _i(ids), ids.size, None)
        return counts, ids

    def pages_to_data_sharded(self, d_pages, n: int, h: int, w: int):
        """Latency mode (ttr_pages_to_data_dev_sharded): rank 0 passes the device pages, the others None."""
        ptr = d_pages.ptr if isinstance(d_pages, DeviceBuffer) else d_pages
        arr = (C.c_void_p * max(n, 1))()
        k = self.lib.ttr_pages_to_data_dev_sharded(self.h, ptr, n, h, w, arr)
        if k < 0:
            raise EngineError(self.lib.ttr_last_error().decode())
        return self.eng._take_many(arr, k) if k else []

    def close(self):
        if getattr(self, "h", None):
            self.lib.ttr_comm_destroy(self.h)
            self.h = None
