"""CPU: the calibrated synthetic PARSeq weights (tuatara_amd/weights.py: _wire_parseq_dfa) decode what their design says, with
top-2 margins far above bf16 noise, and a bf16 noise model of the engine (oracle/bf16sim.py) leaves >= 90 % of the greedy paths
untouched — the property the GPU margin-rule tests (tests/parity_rules.py) rely on."""
import numpy as np
import torch

from tests import parity_rules as R


def _dfa_prediction(parseq, x):
    from tuatara_amd import weights as W
    first, nxt = W.dfa_tables(0)
    with torch.no_grad():
        mem = parseq.encode(x)
    val = (mem[:, :, W._B3:W._B3 + 8] - mem[:, :, W._ZERO:W._ZERO + 1]).mean(1).numpy()
    ref = (mem[:, :, W._B3 + 8] - mem[:, :, W._ZERO]).mean(1).numpy()
    hard = (np.abs(val[:, :6]) > 0.95 * ref[:, None]).all(1)       # every content bit the decoder reads is saturated
    ids = []
    for b in val > 0:
        v = sum(int(b[j]) << j for j in range(6))
        seq = [int(first[v])]
        while len(seq) < 26 and seq[-1] != 0:
            seq.append(int(nxt[seq[-1]]))
        ids.append(seq + [0] * (26 - len(seq)))
    return np.array(ids), hard


def test_designed_readout_decodes_its_tables_with_large_margins(oracle_models):
    _, parseq = oracle_models
    crops = np.random.default_rng(5).integers(0, 256, (64, 32, 128, 3), dtype=np.uint8)
    x = torch.from_numpy(crops).permute(0, 3, 1, 2).float().div(255.0)
    with torch.no_grad():
        ref, ref_ar = parseq(x, return_ar=True)
    ref, ref_ar = ref.numpy(), ref_ar.numpy()
    exp, hard = _dfa_prediction(parseq, x)
    up = R.upto_eos(exp)
    assert hard.mean() > 0.35, hard.mean()                # the rest have a content bit inside the soft knee
    srt = np.sort(ref, -1)
    margin = srt[..., -1] - srt[..., -2]
    for i in np.nonzero(hard)[0]:
        assert np.array_equal(ref.argmax(-1)[i, :up[i]], exp[i, :up[i]]), i          # refined ids = the tables' string
        assert np.array_equal(ref_ar.argmax(-1)[i, :up[i]], exp[i, :up[i]]), i
        assert margin[i, :up[i]].min() > 1.5, (i, margin[i, :up[i]].min())            # bf16 |dlogit| is ~0.05
    from oracle import post
    strs, _ = post.decode_logits(ref)
    assert len(set(strs)) > 24                                                         # the string depends on the crop


def test_bf16_noise_model_keeps_nine_paths_in_ten(oracle_models):
    from oracle import bf16sim
    _, parseq = oracle_models
    crops = np.random.default_rng(0).integers(0, 256, (96, 32, 128, 3), dtype=np.uint8)
    x = torch.from_numpy(crops).permute(0, 3, 1, 2).float().div(255.0)
    with torch.no_grad():
        ref, ref_ar = parseq(x, return_ar=True)
        with bf16sim.bf16_noise():
            got, got_ar = parseq(x, return_ar=True)
        again, _ = parseq(x, return_ar=True)
    assert torch.equal(ref, again)                                                     # the noise model leaves no patch behind
    R.parseq_margin_rule(ref.numpy(), ref_ar.numpy(), got.numpy(), got_ar.numpy(), min_same=0.9, label="bf16 noise model, 96 crops")
