"""Build and run the ONE piece of the reference that compiles here: ``class Tokenizer`` (/root/reference/tuatara.cpp:25-117).

TEST INFRASTRUCTURE ONLY.  Everything else in tuatara.cpp needs OpenCV (absent); the tokenizer depends on LibTorch
and the STL only, and the torch wheel ships LibTorch.  This recipe

  1. extracts the class text from the reference file *where it lies* (never committed: it is written under the git-ignored
     oracle/_ref/ and deleted again after the compile),
  2. compiles it, unmodified, into oracle/_ref/ref_tokenizer with the small driver below (the driver restates the call site
     tuatara.cpp:486-505: softmax, ``decode(pred, false)``, cut at the first EOS character),
  3. with ``--golden`` runs it on id / logit cases (ids 0, 88 and the shifted 69..94 range included) and writes
     tests/golden/g1_ref_tokenizer.json - reference-generated known answers that pin oracle/post.c:orc_decode_ids,
     geometry.cpp:Tokenizer::decode and ttr_decode_ids (tests/test_ref_tokenizer_cpu.py, tests/test_gpu_e2e.py).

  python oracle/build_ref_tokenizer.py [--golden]
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/tuatara.cpp"
OUT = os.path.join(HERE, "_ref")
BIN = os.path.join(OUT, "ref_tokenizer")
GOLDEN = os.path.join(ROOT, "tests", "golden", "g1_ref_tokenizer.json")

DRIVER = r'''
// Driver around the reference's own Tokenizer (included verbatim from the extract): stdin "N L C" + N*L*C floats (logits).
// Per item: torch::softmax(-1) (tuatara.cpp:486), tokenizer.decode(pred, false) (:492), cut at the first EOS char (:497-502).
// Mode "raw": prints decode(pred, true) instead (one char per position, nothing filtered) - used to read the id -> char table.
#include <torch/torch.h>
#include <cstdio>
#include <iostream>
#include <map>
#include <string>
#include <vector>
using namespace torch::indexing;
#include "ref_tokenizer_extract.inc"
int main(int argc, char** argv) {
  bool raw = argc > 1 && std::string(argv[1]) == "raw";
  long N, L, C;
  if (!(std::cin >> N >> L >> C)) return 2;
  std::vector<float> v((size_t)N * L * C);
  for (auto& x : v) std::cin >> x;
  torch::Tensor logits = torch::from_blob(v.data(), {N, L, C}, torch::kFloat).clone();
  auto pred = torch::softmax(logits, -1);
  Tokenizer tokenizer;
  std::vector<std::string> tokens = tokenizer.decode(pred, raw);
  for (auto& t : tokens) {
    std::string out;
    for (char ch : t) {
      if (!raw && ch == tokenizer.EOS) break;
      out.push_back(ch);
    }
    std::printf("%zu", out.size());
    for (unsigned char ch : out) std::printf(" %u", (unsigned)ch);
    std::printf("\n");
  }
  return 0;
}
'''


def extract_class() -> str:
    src = open(REF, encoding="utf-8").read()
    m = re.search(r"^class Tokenizer \{", src, re.M)
    assert m, "class Tokenizer not found in the reference"
    end = src.index("\n};\n", m.start()) + 4
    return src[m.start():end]


def build() -> str | None:
    """Compile oracle/_ref/ref_tokenizer.  Returns its path, or None when the reference is not on this machine."""
    if not os.path.exists(REF):
        return BIN if os.path.exists(BIN) else None
    import torch
    os.makedirs(OUT, exist_ok=True)
    if os.path.exists(BIN) and os.path.getmtime(BIN) > max(os.path.getmtime(REF), os.path.getmtime(__file__)):
        return BIN
    inc = os.path.join(OUT, "ref_tokenizer_extract.inc")
    drv = os.path.join(OUT, "ref_tokenizer_driver.cpp")
    tdir = os.path.dirname(torch.__file__)
    try:
        with open(inc, "w") as f:
            f.write(extract_class())
        with open(drv, "w") as f:
            f.write(DRIVER)
        cmd = ["g++", "-O1", "-std=c++17", drv, "-o", BIN, "-I", os.path.join(tdir, "include"),
               "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"), "-L", os.path.join(tdir, "lib"),
               "-ltorch", "-ltorch_cpu", "-lc10", f"-Wl,-rpath,{os.path.join(tdir, 'lib')}", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}"]
        subprocess.check_call(cmd)
    finally:
        for p in (inc, drv):                      # the extract is reference text: it does not stay on disk
            if os.path.exists(p):
                os.remove(p)
    return BIN


def run(logits, raw: bool = False):
    """logits: nested list / array [N][L][C] -> list of byte strings (as lists of ints) from the reference tokenizer."""
    import numpy as np
    a = np.asarray(logits, np.float32)
    n, l, c = a.shape
    inp = f"{n} {l} {c}\n" + " ".join(repr(float(x)) for x in a.ravel()) + "\n"
    out = subprocess.run([BIN] + (["raw"] if raw else []), input=inp, capture_output=True, text=True, check=True).stdout
    res = []
    for line in out.strip("\n").split("\n"):
        f = line.split()
        res.append([int(x) for x in f[1:1 + int(f[0])]])
    return res


def id_cases():
    import numpy as np
    rng = np.random.default_rng(0)
    cases = [[1, 2, 3, 0, 4], [88, 11, 88, 12, 0], [69, 70, 71, 79, 76, 78, 75, 77, 0], [0], [94, 93, 92, 91, 90, 89, 87, 86],
             [37, 38, 88, 88, 0, 5], list(range(60, 95)), list(range(0, 26)), [88] * 26, list(range(69, 95)),
             [10, 88, 0, 88, 20], [94] * 26]
    for _ in range(12):
        cases.append([int(x) for x in rng.integers(0, 95, 26)])
    return cases


def make_golden() -> dict:
    import numpy as np
    assert build() and os.path.exists(REF), "needs /root/reference"
    rng = np.random.default_rng(1)
    # the id -> char table, read through decode(raw = true) on one-hot distributions over all 98 table entries
    eye = np.full((98, 1, 98), -20.0, np.float32)
    for i in range(98):
        eye[i, 0, i] = 20.0
    itos = [r[0] for r in run(eye, raw=True)]
    kept = run(eye, raw=False)                       # ids whose one-hot decodes to nothing are the ones `filter` drops, or EOS cuts
    dropped = [i for i, r in enumerate(kept) if not r]
    cases = []
    for ids in id_cases():
        lg = rng.normal(0, 1, (1, len(ids), 95)).astype(np.float32)
        for p, t in enumerate(ids):
            lg[0, p, t] = 9.0                        # argmax = the intended id, with realistic noise underneath
        cases.append({"ids": ids, "text": run(lg)[0]})
    g = {"source": "class Tokenizer of /root/reference/tuatara.cpp:25-117 compiled unmodified against the torch wheel's LibTorch; "
                   "call site :486-505 restated in oracle/build_ref_tokenizer.py", "itos": itos, "empty_for_single_id": dropped, "cases": cases}
    with open(GOLDEN, "w") as f:
        json.dump(g, f, indent=0)
    return g


if __name__ == "__main__":
    print(build())
    if "--golden" in sys.argv:
        g = make_golden()
        print("itos:", bytes(g["itos"]).decode("latin1"))
        print("single ids decoding to nothing:", g["empty_for_single_id"], "| cases:", len(g["cases"]))
