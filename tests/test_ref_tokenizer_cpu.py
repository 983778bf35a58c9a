"""CPU: the tokenizer pinned to the REFERENCE ITSELF.  tests/golden/g1_ref_tokenizer.json was produced
by the reference's own `class Tokenizer` (tuatara.cpp:25-117), compiled unmodified against LibTorch by
oracle/build_ref_tokenizer.py; the oracle's restatement (oracle/post.c:orc_decode_ids) and the engine's host decoder
(geometry.cpp:Tokenizer::decode through the C ABI's ttr_decode_ids; no GPU involved) must reproduce it byte for byte."""
import json
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

G = json.load(open(os.path.join(GOLDEN, "g1_ref_tokenizer.json")))


def test_reference_table_has_the_quirks_the_survey_derived():
    itos = bytes(G["itos"]).decode("latin1")
    assert len(itos) == 98 and itos[0] == "]" and itos[88] == "]" and itos[96] == "[" and itos[97] == "P"
    assert itos[69:79] == "\\'()*+,-./"                      # ids >= 69 shifted by one against upstream PARSeq
    assert G["empty_for_single_id"] == [0, 88]               # 88 is filtered as eos_id, 0 decodes to ']' and is cut


def test_oracle_restatement_matches_the_reference_tokenizer():
    from oracle import post
    itos, eos, bos, pad = post.tokenizer_table()
    assert [ord(c) for c in itos] == G["itos"] and eos == 88
    for c in G["cases"]:
        assert [ord(ch) for ch in post.decode_ids(c["ids"])] == c["text"], c["ids"]


def test_engine_host_decoder_matches_the_reference_tokenizer():
    from tuatara_amd.engine import decode_ids
    for c in G["cases"]:
        assert [ord(ch) for ch in decode_ids(c["ids"])] == c["text"], c["ids"]


def test_committed_vectors_are_what_the_reference_produces_here():
    """Where the reference is on this machine (the build container), rebuild its tokenizer and regenerate the cases."""
    from oracle import build_ref_tokenizer as B
    if not os.path.exists(B.REF):
        pytest.skip("no /root/reference on this machine")
    assert B.build()
    rng = np.random.default_rng(1)
    eye = np.full((98, 1, 98), -20.0, np.float32)
    for i in range(98):
        eye[i, 0, i] = 20.0
    assert [r[0] for r in B.run(eye, raw=True)] == G["itos"]
    lg = rng.normal(0, 1, (1, 26, 95)).astype(np.float32)
    ids = [int(x) for x in lg[0].argmax(-1)]
    from oracle import post
    assert B.run(lg)[0] == [ord(ch) for ch in post.decode_ids(ids)]
