"""CPU: tools/convert_weights.py.  The reference's two TorchScript archives are not obtainable here, so
the converter is exercised on archives of the same layout made from the oracle models: trace -> save under the reference's
file names (tuatara.cpp:333, :423) -> convert -> the .ttrw files must equal the ones written directly from the state dicts."""
import os
import subprocess
import sys

import numpy as np
import torch

from tests.conftest import ROOT


def test_converter_round_trip(tmp_path, weights, oracle_models):
    from tuatara_amd import weights as W
    craft, parseq = oracle_models
    d = str(tmp_path)
    with torch.no_grad():
        tc = torch.jit.trace(craft, torch.zeros(1, 3, 64, 64), check_trace=False)
        tp = torch.jit.trace(parseq, torch.zeros(1, 3, 32, 128), check_trace=False)
    tc.save(os.path.join(d, "craft_traced_torchscript_model.pt"))
    tp.save(os.path.join(d, "parseq_torchscript.bin"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), d], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    for name in (W.CRAFT_FILE, W.PARSEQ_FILE):
        got, ref = W.read_ttrw(os.path.join(d, name)), W.read_ttrw(os.path.join(weights["dir"], name))
        assert got.keys() == ref.keys()
        for k in ref:
            assert np.array_equal(got[k], ref[k]), (name, k)


def test_converter_rejects_foreign_archive(tmp_path):
    d = str(tmp_path)
    m = torch.jit.trace(torch.nn.Conv2d(3, 4, 3), torch.zeros(1, 3, 8, 8))
    m.save(os.path.join(d, "craft_traced_torchscript_model.pt"))
    m.save(os.path.join(d, "parseq_torchscript.bin"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), d], capture_output=True, text=True)
    assert out.returncode != 0 and "does not match the expected architecture" in (out.stderr + out.stdout)
