"""CPU: the example CLI's PNG reader (examples/png_decode.h) against PIL for every PNG flavour it claims,
and the run_ocr counterpart's annotation helper.  The reference's CLIs read with cv::imread(IMREAD_COLOR)
(examples/resume.cpp:9): BGR, alpha dropped, grey replicated."""
import os
import subprocess
import sys

import numpy as np
import pytest
from PIL import Image

from tests.conftest import DATA, ROOT


@pytest.fixture(scope="module")
def cli():
    from tuatara_amd import build as B
    B.build_lib()
    return B.build_examples()


def _decode(cli, path, tmp_path):
    raw = str(tmp_path / "out.raw")
    out = subprocess.run([cli, "--decode-only", path, raw], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rows, cols = map(int, out.stdout.split())
    return np.fromfile(raw, np.uint8).reshape(rows, cols, 3)


def test_png_reader_funsd_fixture(cli, tmp_path):
    got = _decode(cli, os.path.join(DATA, "funsd_0001129658.png"), tmp_path)
    ref = np.array(Image.open(os.path.join(DATA, "funsd_0001129658.png")).convert("RGB"))[:, :, ::-1]
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("mode", ["L", "RGB", "RGBA", "LA", "P", "1", "I;16"])
def test_png_reader_flavours(cli, tmp_path, mode):
    rng = np.random.default_rng(5)
    h, w = 37, 53                                   # odd sizes: sub-byte rows end mid-byte
    if mode == "I;16":
        img = Image.fromarray(rng.integers(0, 65536, (h, w), dtype=np.uint16))
        ref = (np.array(img) >> 8).astype(np.uint8)
        ref = np.repeat(ref[:, :, None], 3, 2)
    elif mode == "1":
        img = Image.fromarray(rng.integers(0, 2, (h, w), dtype=np.uint8) * 255).convert("1")
        ref = np.array(img.convert("RGB"))[:, :, ::-1]
    elif mode == "P":
        img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).convert("P", palette=Image.ADAPTIVE, colors=200)
        ref = np.array(img.convert("RGB"))[:, :, ::-1]
    else:
        nch = {"L": 1, "RGB": 3, "RGBA": 4, "LA": 2}[mode]
        a = rng.integers(0, 256, (h, w, nch), dtype=np.uint8)
        img = Image.fromarray(a[:, :, 0] if nch == 1 else a, mode)
        rgb = np.repeat(a[:, :, :1], 3, 2) if nch <= 2 else a[:, :, :3]   # alpha dropped, not composited (cv::imread)
        ref = rgb[:, :, ::-1]
    path = str(tmp_path / f"t_{mode.replace(';', '')}.png")
    img.save(path)
    assert np.array_equal(_decode(cli, path, tmp_path), ref)


def test_cli_rejects_garbage(cli, tmp_path):
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not a png at all")
    out = subprocess.run([cli, "--decode-only", str(bad), str(tmp_path / "o.raw")], capture_output=True, text=True)
    assert out.returncode == 1 and "not a PNG" in out.stderr


def test_run_ocr_annotate_draws_the_three_panels():
    """The reference's draw_boxes_and_text (run_ocr.py:10-82): page with boxes | texts at their boxes | running text sorted by (y1, x1)."""
    sys.path.insert(0, os.path.join(ROOT, "bindings"))
    import run_ocr
    img = np.full((60, 100, 3), 255, np.uint8)
    res = [{"text": "cd", "bbox": [50.0, 30.0, 90.0, 44.0]}, {"text": "ab", "bbox": [10.0, 10.0, 40.0, 22.0]}]
    out = np.array(run_ocr.annotate(img, res))
    assert out.shape == (60, 300, 3)
    assert (out[10, 10:40] != 255).any() and (out[30, 50:90] != 255).any()          # green boxes on the page copy
    assert tuple(out[10, 20]) == (0, 255, 0)
    assert (out[10:22, 110:140] != 0).any() and (out[30:44, 150:190] != 0).any()      # each text at its box, second panel
    third = out[:, 200:]
    ys, xs = np.nonzero(third.any(-1))
    assert len(xs) and xs.min() >= 10 and ys.max() <= 32                              # running text starts at (10, 30), third panel
    first_word = third[:, :10 + (xs.max() - 10) // 2]
    assert first_word.any()                                                           # "ab" (smaller y1) comes first on the line
