import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "tests", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu() -> bool:
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


@pytest.fixture(scope="session")
def weights(tmp_path_factory):
    """Synthetic seeded weights: raw state dicts + the two .ttrw files (structured CRAFT)."""
    from tuatara_amd import weights as W

    d = str(tmp_path_factory.mktemp("weights"))
    c, p = W.make_synthetic_weights(d, seed=0, structured=True)
    return {"dir": d, "craft": c, "parseq": p}


@pytest.fixture(scope="session")
def weights_random(tmp_path_factory):
    """Fully random CRAFT weights (every layer matters for the heat map) + the same PARSeq."""
    from tuatara_amd import weights as W

    d = str(tmp_path_factory.mktemp("weights_rand"))
    c, p = W.make_synthetic_weights(d, seed=0, structured=False)
    return {"dir": d, "craft": c, "parseq": p}


@pytest.fixture(scope="session")
def oracle_models(weights):
    from oracle import pipeline

    return pipeline.load_models(weights["craft"], weights["parseq"])


@pytest.fixture(scope="session")
def funsd():
    from PIL import Image

    return np.array(Image.open(os.path.join(DATA, "funsd_0001129658.png")).convert("RGB"))


@pytest.fixture(scope="session")
def funsd_oracle(oracle_models, funsd):
    """The CPU oracle on the FUNSD page (config 1 / 4), once per session: detect()'s fields + crops, logits, result (oracle/pipeline.py, debug=True)."""
    from oracle import pipeline
    return pipeline.image_to_data(oracle_models[0], oracle_models[1], funsd, debug=True)


def _engine(weights_dir, precision):
    from tuatara_amd.build import build_lib
    from tuatara_amd.engine import Engine

    build_lib()
    return Engine(weights_dir, precision=precision)


@pytest.fixture(scope="session")
def eng_f32(weights):
    return _engine(weights["dir"], "f32")


@pytest.fixture(scope="session")
def eng_x4(weights):
    """the engine in its DEFAULT precision (f16x4: fp32-equivalent split-operand f16 MFMA)"""
    return _engine(weights["dir"], "f16x4")


@pytest.fixture(scope="session")
def eng_bf16(weights):
    return _engine(weights["dir"], "bf16")
