#!/usr/bin/env python3
"""Convert the reference's TorchScript archives to the engine's flat weight files.

  python tools/convert_weights.py <weights_dir>

Reads <weights_dir>/craft_traced_torchscript_model.pt (tuatara.cpp:333) and
<weights_dir>/parseq_torchscript.bin (tuatara.cpp:423) with torch.jit.load (CPU), takes their
state_dict (upstream CRAFT / PARSeq parameter names), folds BatchNorm and writes craft.ttrw /
parseq.ttrw next to them.  The real archives are not obtainable offline; tests/test_convert_cpu.py runs this script on
archives of the same layout traced from the oracle models and requires bit-identical .ttrw files.  The key/shape check
below fails loudly if an archive does not match the upstream architectures.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _state(path):
    import torch

    m = torch.jit.load(path, map_location="cpu")
    sd = {k: v.detach().float().numpy() for k, v in m.state_dict().items()}
    # traced modules sometimes prefix everything with "model." / "module."
    for pre in ("model.", "module."):
        if sd and all(k.startswith(pre) for k in sd):
            sd = {k[len(pre):]: v for k, v in sd.items()}
    return sd


def _check(sd, spec, what):
    missing = [k for k, _ in spec if k not in sd]
    wrong = [(k, sd[k].shape, s) for k, s in spec if k in sd and tuple(sd[k].shape) != tuple(s)]
    if missing or wrong:
        raise SystemExit(f"{what}: archive does not match the expected architecture; missing={missing[:5]} wrong={wrong[:5]}")


def main():
    from tuatara_amd import weights as W

    d = sys.argv[1]
    c = _state(os.path.join(d, "craft_traced_torchscript_model.pt"))
    _check(c, W.craft_spec(), "CRAFT")
    p = _state(os.path.join(d, "parseq_torchscript.bin"))
    _check(p, W.parseq_spec(), "PARSeq")
    print(W.export_craft(c, d), W.export_parseq(p, d))


if __name__ == "__main__":
    main()
