"""Build and run the reference's own recogniser fan-out: ``infer`` (/root/reference/tuatara.cpp:289-312) + ``class Tokenizer`` (:25-117).

TEST INFRASTRUCTURE ONLY.  Besides the tokenizer, ``infer`` is the one other piece of tuatara.cpp that depends on LibTorch and the
STL alone (everything else needs OpenCV, absent here), and the torch wheel ships LibTorch.  This recipe

  1. extracts the two texts from the reference file *where it lies* (written under the git-ignored oracle/_ref/, deleted again after
     the compile; never committed),
  2. compiles them, unmodified, into oracle/_ref/ref_infer with the driver below.  The driver restates the call site
     tuatara.cpp:423-428 (torch::jit::load of parseq_torchscript.bin), :443-446 (from_blob {1, 32, 128, 3} kByte -> permute -> kFloat
     -> div 255; the crops arrive already resized and channel-swapped, the OpenCV half :440-441), :450-459 (chunks of 4 into the
     queue), :461-475 (6 threads running the reference's ``infer`` on ONE shared module), :478-486 (sort by first index, cat,
     softmax), :491-505 (Tokenizer::decode, cut at the first EOS character),
  3. with ``--golden`` exports the oracle's PARSeq (seed-0 synthetic weights) as a TorchScript archive the way the reference's archive is
     consumed, runs the binary on 22 seeded crops (five chunks of 4 and a ragged one of 2) and writes tests/golden/g8_ref_infer.npz:
     crops, the [22, 26, 95] logits and the strings the REFERENCE-COMPILED code produced.  That pins rows a9 (chunking / threads /
     sort / cat) and the LibTorch-C++ execution leg of a10, plus a11 end to end, for the oracle (tests/test_ref_infer_cpu.py) and the
     engine (tests/test_gpu_x4_parity.py::test_x4_matches_reference_compiled_infer).

What it cannot pin: the reference's real weights (not in the tree, no network) - the graph is the oracle's restatement of upstream
PARSeq with synthetic weights; the execution, batching, threading, softmax and decoding are the reference's own compiled code.

  python oracle/build_ref_infer.py [--golden]
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF = "/root/reference/tuatara.cpp"
OUT = os.path.join(HERE, "_ref")
BIN = os.path.join(OUT, "ref_infer")
GOLDEN = os.path.join(ROOT, "tests", "golden", "g8_ref_infer.npz")
N_GOLDEN, SEED_GOLDEN = 22, 8

DRIVER = r'''
// Driver around the reference's own infer() and Tokenizer (included verbatim from the extract).
//   ref_infer <parseq_torchscript.bin> <crops.u8: N x 32 x 128 x 3> <N> <logits.f32 out>     stdout: one line per crop, "len b0 b1 ..."
#include <torch/script.h>
#include <torch/torch.h>
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>
using namespace torch::indexing;
#include "ref_infer_extract.inc"
int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const long N = std::atol(argv[3]);
  std::vector<unsigned char> crops((size_t)N * 32 * 128 * 3);
  { std::ifstream f(argv[2], std::ios::binary); if (!f.read((char*)crops.data(), (std::streamsize)crops.size())) return 3; }
  torch::jit::script::Module parseq_model;                                       // tuatara.cpp:426-432
  try { parseq_model = torch::jit::load(argv[1]); } catch (const c10::Error& e) { std::cerr << "error loading the parseq model\n"; return 4; }
  std::vector<torch::Tensor> parseq_tensors;                                     // :436-448 (the resize / cvtColor of :440-441 happened upstream of this file)
  for (long i = 0; i < N; ++i) {
    torch::Tensor parseq_tensor = torch::from_blob(crops.data() + (size_t)i * 32 * 128 * 3, {1, 32, 128, 3}, torch::kByte);
    parseq_tensor = parseq_tensor.permute({0, 3, 1, 2});
    parseq_tensor = parseq_tensor.to(torch::kFloat);
    parseq_tensor = parseq_tensor.div(255.0);
    parseq_tensors.push_back(parseq_tensor);
  }
  std::queue<std::pair<int, torch::Tensor>> input_queue;                          // :450-459
  int chunk_size = 4;
  for (size_t i = 0; i < parseq_tensors.size(); i += chunk_size) {
    std::vector<torch::Tensor> chunk(parseq_tensors.begin() + i, parseq_tensors.begin() + std::min(i + chunk_size, parseq_tensors.size()));
    input_queue.push(std::make_pair(i, torch::cat(chunk, 0)));
  }
  const int num_threads = 6;                                                      // :461-475
  std::vector<std::thread> threads;
  std::vector<std::pair<int, torch::Tensor>> parseq_outputs;
  std::mutex input_mutex, output_mutex;
  for (int i = 0; i < num_threads; i++)
    threads.emplace_back(infer, std::ref(parseq_model), std::ref(input_queue), std::ref(parseq_outputs), std::ref(input_mutex), std::ref(output_mutex));
  for (auto& thread : threads) if (thread.joinable()) thread.join();
  std::sort(parseq_outputs.begin(), parseq_outputs.end(), [](const std::pair<int, torch::Tensor>& a, const std::pair<int, torch::Tensor>& b) { return a.first < b.first; });   // :478
  std::vector<torch::Tensor> sorted_outputs;
  for (const auto& output : parseq_outputs) sorted_outputs.push_back(output.second);
  torch::Tensor parseq_output_tensor = torch::cat(sorted_outputs, 0).contiguous();   // :485
  { std::ofstream f(argv[4], std::ios::binary); f.write((const char*)parseq_output_tensor.data_ptr<float>(), (std::streamsize)(parseq_output_tensor.numel() * 4)); }
  auto parseq_pred = torch::softmax(parseq_output_tensor, -1);                    // :486
  Tokenizer tokenizer;                                                            // :491-505
  std::vector<std::string> tokens = tokenizer.decode(parseq_pred, false);
  for (auto& t : tokens) {
    std::string predicted_text;
    for (const auto& token_char : t) { if (token_char == tokenizer.EOS) break; predicted_text.push_back(token_char); }
    std::printf("%zu", predicted_text.size());
    for (unsigned char ch : predicted_text) std::printf(" %u", (unsigned)ch);
    std::printf("\n");
  }
  std::fprintf(stderr, "ref_infer: %ld crops, %zu chunks, output %s\n", N, parseq_outputs.size(), std::to_string(parseq_output_tensor.size(0)).c_str());
  return 0;
}
'''


def extract() -> str:
    src = open(REF, encoding="utf-8").read()
    m = re.search(r"^class Tokenizer \{", src, re.M)
    assert m, "class Tokenizer not found in the reference"
    tok = src[m.start():src.index("\n};\n", m.start()) + 4]
    m = re.search(r"^void infer\(", src, re.M)
    assert m, "infer() not found in the reference"
    inf = src[m.start():src.index("\n}\n", m.start()) + 3]
    return tok + "\n" + inf


def build() -> str | None:
    """Compile oracle/_ref/ref_infer.  Returns its path, or None when the reference is not on this machine (and no earlier build is)."""
    if not os.path.exists(REF):
        return BIN if os.path.exists(BIN) else None
    import torch
    os.makedirs(OUT, exist_ok=True)
    if os.path.exists(BIN) and os.path.getmtime(BIN) > max(os.path.getmtime(REF), os.path.getmtime(__file__)):
        return BIN
    inc = os.path.join(OUT, "ref_infer_extract.inc")
    drv = os.path.join(OUT, "ref_infer_driver.cpp")
    tdir = os.path.dirname(torch.__file__)
    try:
        with open(inc, "w") as f:
            f.write(extract())
        with open(drv, "w") as f:
            f.write(DRIVER)
        cmd = ["g++", "-O1", "-std=c++17", drv, "-o", BIN, "-I", os.path.join(tdir, "include"),
               "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"), "-L", os.path.join(tdir, "lib"),
               "-ltorch", "-ltorch_cpu", "-lc10", "-lpthread", f"-Wl,-rpath,{os.path.join(tdir, 'lib')}",
               f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}"]
        subprocess.check_call(cmd)
    finally:
        for p in (inc, drv):                      # the extract is reference text: it does not stay on disk
            if os.path.exists(p):
                os.remove(p)
    return BIN


def export_parseq_archive(path: str, seed: int = 0):
    """The oracle's PARSeq with the seeded synthetic weights as the TorchScript archive the reference loads (tuatara.cpp:423)."""
    import warnings

    import torch

    from oracle import pipeline
    from tuatara_amd import weights as W
    _, parseq = pipeline.load_models(W.synth_craft(seed, True), W.synth_parseq(seed))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.jit.trace(parseq, torch.zeros(4, 3, 32, 128), check_trace=False).save(path)   # (the trace is batch-generic: sizes are traced)
    return parseq


def run(archive: str, crops):
    """crops u8 [N, 32, 128, 3] -> (logits f32 [N, 26, 95], list of byte-lists) from the reference-compiled fan-out."""
    import numpy as np
    crops = np.ascontiguousarray(crops, np.uint8)
    n = crops.shape[0]
    with tempfile.TemporaryDirectory() as d:
        cpath, lpath = os.path.join(d, "crops.u8"), os.path.join(d, "logits.f32")
        crops.tofile(cpath)
        out = subprocess.run([BIN, archive, cpath, str(n), lpath], capture_output=True, text=True, check=True).stdout
        logits = np.fromfile(lpath, np.float32).reshape(n, 26, 95)
    texts = []
    for line in out.strip("\n").split("\n"):
        f = line.split()
        texts.append([int(x) for x in f[1:1 + int(f[0])]])
    assert len(texts) == n
    return logits, texts


def golden_crops():
    """22 crops: half uniform noise, half 'text-like' (dark strokes on a light ground), so that the strings differ in length."""
    import numpy as np
    rng = np.random.default_rng(SEED_GOLDEN)
    crops = rng.integers(0, 256, (N_GOLDEN, 32, 128, 3), dtype=np.uint8)
    for i in range(N_GOLDEN // 2, N_GOLDEN):
        img = np.full((32, 128, 3), 235, np.uint8)
        for _ in range(int(rng.integers(2, 9))):
            x, w = int(rng.integers(2, 118)), int(rng.integers(2, 9))
            y, h = int(rng.integers(3, 14)), int(rng.integers(8, 18))
            img[y:y + h, x:x + w] = rng.integers(0, 90, (1, 1, 3), dtype=np.uint8)
        crops[i] = img
    return crops


def make_golden() -> dict:
    import numpy as np
    assert build() and os.path.exists(REF), "needs /root/reference"
    crops = golden_crops()
    with tempfile.TemporaryDirectory() as d:
        arch = os.path.join(d, "parseq_torchscript.bin")
        export_parseq_archive(arch)
        logits, texts = run(arch, crops)
        logits2, texts2 = run(arch, crops)                       # six threads on one module: the result must not depend on the schedule
    assert np.array_equal(logits, logits2) and texts == texts2, "the reference fan-out is not deterministic here"
    np.savez_compressed(GOLDEN, crops=crops, logits=logits, texts=np.frombuffer(json.dumps(texts).encode(), np.uint8),
                        source=np.frombuffer(("infer() and class Tokenizer of /root/reference/tuatara.cpp (:289-312, :25-117) compiled unmodified against the torch "
                                              "wheel's LibTorch; call site :423-505 restated in oracle/build_ref_infer.py; module = oracle PARSeq, seed-0 synthetic "
                                              "weights, exported with torch.jit.trace").encode(), np.uint8))
    return {"logits": logits, "texts": texts}


if __name__ == "__main__":
    print(build())
    if "--golden" in sys.argv:
        g = make_golden()
        print("strings:", [bytes(t).decode("latin1") for t in g["texts"]])
        print("max |logit|:", float(abs(g["logits"]).max()))
