"""CPU suite: the oracle's PARSeq + batching + decoding against what the REFERENCE's own compiled code produced.

tests/golden/g8_ref_infer.npz (oracle/build_ref_infer.py --golden): infer() (/root/reference/tuatara.cpp:289-312) and class Tokenizer (:25-117)
compiled unmodified against LibTorch, driven the way tuatara.cpp:423-505 drives them (chunks of 4, six threads on one module, sort, cat, softmax,
decode, EOS cut) on the TorchScript export of the seed-0 PARSeq.  Pins SURVEY.md section 8 rows a9 / a10 (LibTorch-C++ execution leg) / a11 for the
oracle here and for the engine in tests/test_gpu_x4_parity.py::test_x4_matches_reference_compiled_infer."""
import json
import os

import numpy as np

from tests.conftest import GOLDEN


def _golden():
    g = np.load(os.path.join(GOLDEN, "g8_ref_infer.npz"))
    texts = [bytes(t).decode("latin1") for t in json.loads(bytes(g["texts"]).decode())]
    return g["crops"], g["logits"], texts


def test_fixture_shape_and_provenance():
    g = np.load(os.path.join(GOLDEN, "g8_ref_infer.npz"))
    assert g["crops"].shape == (22, 32, 128, 3) and g["crops"].dtype == np.uint8      # five chunks of 4 and a ragged one of 2 (tuatara.cpp:452-458)
    assert g["logits"].shape == (22, 26, 95) and g["logits"].dtype == np.float32
    assert "tuatara.cpp" in bytes(g["source"]).decode() and "compiled unmodified" in bytes(g["source"]).decode()
    _, _, texts = _golden()
    assert len(texts) == 22 and any(len(t) >= 8 for t in texts) and any(t == "" for t in texts)


def test_oracle_one_batch_equals_reference_fanout(oracle_models):
    """The oracle runs all crops as one eager batch; the reference ran them as 4-crop chunks on six threads through the TorchScript interpreter.  Same
    logits to fp32 summation noise (SURVEY: batch composition moves a logit by <= 1.1e-5), the same ids at all 26 positions, the same strings through
    the oracle's decoder (oracle/post.c) and the engine's host decoder (geometry.cpp via ttr_decode_ids)."""
    from oracle import pipeline, post
    from tuatara_amd.engine import decode_ids
    crops, ref, texts = _golden()
    got = pipeline.parseq_logits(oracle_models[1], crops)
    chunked = np.concatenate([pipeline.parseq_logits(oracle_models[1], crops[i:i + 4]) for i in range(0, len(crops), 4)])   # the reference's batching, eagerly
    err, err4 = np.abs(got - ref).max(), np.abs(chunked - ref).max()
    print(f"oracle vs reference-compiled infer(): one eager batch max |dlogit| {err:.2e}; eager chunks of 4 {err4:.2e}" + (" (bit-equal)" if err4 == 0 else ""))
    # measured in the build container: chunks of 4 reproduce the reference's logits BIT FOR BIT (the eager modules and the TorchScript interpreter run the
    # same ATen kernels on the same shapes); one batch of 22 differs by 2e-4 (other GEMM blocking: fp32 summation order - the fp32 evaluation itself
    # sits 4.5e-4 from an fp64 one on these crops).  Another CPU may pick other kernels: the bound is the noise level, the print says which case this is
    assert err4 < 5e-4 and err < 5e-4
    assert np.array_equal(got.argmax(-1), ref.argmax(-1)) and np.array_equal(chunked.argmax(-1), ref.argmax(-1))
    s, ids = post.decode_logits(ref)
    assert s == texts
    assert [decode_ids(r) for r in ref.argmax(-1)] == texts


def test_reference_binary_reproduces_the_fixture_when_the_reference_is_here(tmp_path):
    """In the build container (/root/reference present) the recipe runs again: same strings, logits within thread-schedule / CPU-kernel noise of the
    committed ones.  Skipped on machines without the reference (the GPU box)."""
    import pytest
    from oracle import build_ref_infer as B
    if not os.path.exists(B.REF):
        pytest.skip("no /root/reference here")
    assert B.build()
    arch = str(tmp_path / "parseq_torchscript.bin")
    B.export_parseq_archive(arch)
    crops, ref, texts = _golden()
    assert np.array_equal(B.golden_crops(), crops)
    logits, t = B.run(arch, crops)
    assert np.abs(logits - ref).max() < 5e-4
    assert [bytes(x).decode("latin1") for x in t] == texts
