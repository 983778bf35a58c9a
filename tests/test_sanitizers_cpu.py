"""CPU suite: the engine's host-only code under sanitizers.  tests/native/host_san.cpp links geometry.cpp with the
.ttrw reader and the host-thread pool (tuatara_amd/csrc/host_util.h) and the example CLI's PNG reader (examples/png_decode.h) - no HIP -
once with -fsanitize=address,undefined and once with -fsanitize=thread.  Truncated and corrupted weight files and PNGs must be REJECTED
with a C++ exception (exit code 0, "rejected: ..."): any out-of-bounds read, overflow or data race turns into a sanitizer report and a
non-zero exit.  The pool replaces the reference's ad-hoc thread fan-out (tuatara.cpp:461-475)."""
import os
import shutil
import struct
import subprocess
import zlib

import numpy as np
import pytest

from tests.conftest import ROOT

SRC = [os.path.join(ROOT, "tests", "native", "host_san.cpp"), os.path.join(ROOT, "tuatara_amd", "csrc", "geometry.cpp")]


def _build(tmp, name, flags):
    out = os.path.join(tmp, name)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer"] + flags + SRC + ["-o", out, "-lz", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip(f"sanitizer build not available here: {r.stderr[-400:]}")
    return out


@pytest.fixture(scope="module")
def san_bins(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    tmp = str(tmp_path_factory.mktemp("san"))
    return {"asan": _build(tmp, "host_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
            "tsan": _build(tmp, "host_tsan", ["-fsanitize=thread"])}


def _run(binary, *args, timeout=300):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([binary, *[str(a) for a in args]], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (args, r.stdout[-500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r.stdout.strip()


def _png(w=37, h=21, ctype=2, depth=8):
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    rng = np.random.default_rng(1)
    stride = (w * ch * depth + 7) // 8
    raw = b"".join(bytes([y % 5]) + rng.integers(0, 256, stride, dtype=np.uint8).tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    plte = chunk(b"PLTE", bytes(range(256)) * 3) if ctype == 3 else b""
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + plte + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def test_weight_file_reader_rejects_hostile_files(san_bins, tmp_path, weights):
    good = os.path.join(weights["dir"], "parseq.ttrw")
    assert _run(san_bins["asan"], "ttrw", good).startswith("ok ")
    data = open(good, "rb").read()
    rng = np.random.default_rng(0)
    cases = {"empty": b"", "magic_only": data[:8], "short_header": data[:40], "half": data[: len(data) // 2], "minus_one": data[:-1],
             "count_huge": data[:8] + struct.pack("<I", 0xFFFFFFFF) + data[12:4096]}
    head = bytearray(data[:20000])
    for k in range(24):                                     # random byte flips inside the tensor table (names, dims, offsets, sizes)
        h = bytearray(head)
        for _ in range(1 + k % 4):
            pos = int(rng.integers(8, 6000))
            h[pos] = int(rng.integers(0, 256))
        cases[f"flip{k}"] = bytes(h) + data[20000:200000]
    n_rejected = 0
    for name, blob in cases.items():
        p = tmp_path / f"{name}.ttrw"
        p.write_bytes(blob)
        out = _run(san_bins["asan"], "ttrw", p)
        assert out.startswith("ok ") or out.startswith("rejected: "), (name, out)
        n_rejected += out.startswith("rejected")
    assert n_rejected >= 8, n_rejected
    assert _run(san_bins["asan"], "ttrw", tmp_path / "does_not_exist.ttrw").startswith("rejected: cannot open")


def test_png_reader_rejects_hostile_files(san_bins, tmp_path, funsd):
    good = os.path.join(ROOT, "tests", "data", "funsd_0001129658.png")
    assert _run(san_bins["asan"], "png", good) == f"ok {funsd.shape[1]}x{funsd.shape[0]}"
    for ctype, depth in ((0, 1), (0, 8), (0, 16), (2, 8), (2, 16), (3, 4), (3, 8), (4, 8), (6, 8), (6, 16)):
        p = tmp_path / f"ok_{ctype}_{depth}.png"
        p.write_bytes(_png(37, 21, ctype, depth))
        assert _run(san_bins["asan"], "png", p) == "ok 37x21", (ctype, depth)
    data = _png(64, 48, 2, 8)
    rng = np.random.default_rng(2)
    cases = {"empty": b"", "sig": data[:8], "ihdr_cut": data[:20], "no_idat": data[:33] + data[-12:], "half": data[: len(data) // 2],
             "huge_dims": data[:16] + struct.pack(">II", 65535, 65535) + data[24:], "zero_dims": data[:16] + struct.pack(">II", 0, 7) + data[24:],
             "bad_depth": data[:24] + b"\x07" + data[25:], "bad_ctype": data[:25] + b"\x05" + data[26:], "interlaced": data[:28] + b"\x01" + data[29:],
             "len_overflow": data[:33] + struct.pack(">I", 0xFFFFFFF0) + data[37:]}
    for k in range(30):
        d = bytearray(data)
        for _ in range(1 + k % 5):
            d[int(rng.integers(8, len(d)))] = int(rng.integers(0, 256))
        cases[f"flip{k}"] = bytes(d)
    n_rejected = 0
    for name, blob in cases.items():
        p = tmp_path / f"{name}.png"
        p.write_bytes(blob)
        out = _run(san_bins["asan"], "png", p)
        assert out.startswith("ok ") or out.startswith("rejected: "), (name, out)
        n_rejected += out.startswith("rejected")
    assert n_rejected >= 10, n_rejected


def test_geometry_and_tokenizer_under_asan_ubsan(san_bins):
    for seed in (1, 2, 3):
        assert _run(san_bins["asan"], "geom", seed, 300).startswith("ok geom")


@pytest.mark.parametrize("threads", [1, 3, 15])
def test_host_pool_under_tsan_and_asan(san_bins, threads):
    """Uneven tasks, exceptions thrown inside tasks (the first one is rethrown to the caller), reuse of one pool for hundreds of
    batches, construction and teardown - no data race (TSan), no leak or use-after-free (ASan)."""
    assert _run(san_bins["tsan"], "pool", threads, 150).startswith("ok pool")
    assert _run(san_bins["asan"], "pool", threads, 150).startswith("ok pool")
