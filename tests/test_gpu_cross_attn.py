"""-m gpu: the decoder's cross-attention kernels of the default precision on their own (ttr_dbg_cross_attn), against a float64 evaluation.

nn.MultiheadAttention of the decoder layer inside the TorchScript PARSeq the reference runs at /root/reference/tuatara.cpp:307: 12 heads of 32, R query rows
of a crop against its 128 memory tokens.  Three kernels serve it: on the matrix cores in split-operand arithmetic (attn_cross_split.hip: the refinement pass,
R <= 32), one workgroup per crop on the vector ALU (R <= 26) and one per row with the heads in 1 - 12 workgroups (the AR steps: R = 1).  Every form must be an
fp32-grade evaluation of softmax(q k^T / sqrt(32)) v: the test states the error against float64 in units of max |v|."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ref(q, kv):
    N, R = q.shape[:2]
    q = q.astype(np.float64).reshape(N, R, 12, 32)
    k = kv[..., :384].astype(np.float64).reshape(N, 128, 12, 32)
    v = kv[..., 384:].astype(np.float64).reshape(N, 128, 12, 32)
    s = np.einsum("nrhd,njhd->nhrj", q, k) / np.sqrt(32.0)
    s -= s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return np.einsum("nhrj,njhd->nrhd", p, v).reshape(N, R, 384)


def _inputs(N, R, seed, sharp=1.0):
    rng = np.random.default_rng(seed)
    q = (rng.standard_normal((N, R, 384)) * sharp).astype(np.float32)
    kv = rng.standard_normal((N, 128, 768)).astype(np.float32)
    kv[..., 384:] *= np.float32(3.0)                       # values of a few units, like the memory's
    return q, kv


@pytest.fixture
def knobs(eng_x4):
    def set_(**kw):
        for k, v in kw.items():
            assert eng_x4.set_tuning(k.encode(), v) == 0, k
    yield set_
    set_(cross_split=1, cross_crop=1, cross_rows_hsplit=4)


# (N, R): the refinement pass's 26 rows at a ragged and a single crop count, fewer rows, the 32-row limit of the matrix-core kernel, the AR steps' single row
CASES = [(37, 26), (1, 26), (5, 7), (3, 32), (52, 1), (200, 1)]


@pytest.mark.parametrize("case", CASES)
def test_cross_attention_kernels_against_float64(eng_x4, knobs, case):
    N, R = case
    for sharp, seed in ((1.0, 1), (6.0, 2)):                # scores of a few units, and near one-hot rows (|score| ~ 30)
        q, kv = _inputs(N, R, seed, sharp)
        ref = _ref(q, kv)
        vmax = float(np.abs(kv[..., 384:]).max())
        forms = {"matrix cores": dict(cross_split=1, cross_crop=1), "per crop": dict(cross_split=0, cross_crop=1),
                 "per row": dict(cross_split=0, cross_crop=0, cross_rows_hsplit=1), "per row, head groups": dict(cross_split=0, cross_crop=0, cross_rows_hsplit=4)}
        got = {}
        for name, kn in forms.items():
            knobs(**kn)
            got[name] = eng_x4.dbg_cross_attn(q, kv)
            err = float(np.abs(got[name] - ref).max()) / vmax
            print(f"N={N} R={R} sharp={sharp}: {name}: max |err| / max |v| = {err:.2e}")
            # fp32 evaluation: 128 products of p <= 1 and |v| <= vmax; the split form adds the 2^-22 of its pairs (p, v) - same order
            assert np.isfinite(got[name]).all() and err < 2e-6, (name, N, R, sharp, err)
        # the two per-row forms compute every head with the same instructions in the same order
        assert np.array_equal(got["per row"], got["per row, head groups"])


def test_cross_attention_rows_behind_the_26th_are_not_written(eng_x4, knobs):
    """The matrix-core kernel pads a crop's queries to 32 rows in registers; rows R .. 31 must neither be read nor stored (the next crop's rows follow
    directly in memory): 3 crops of 7 rows, every output row equal to the one-crop call of its crop."""
    knobs(cross_split=1, cross_crop=1)
    q, kv = _inputs(3, 7, 9)
    whole = eng_x4.dbg_cross_attn(q, kv)
    for n in range(3):
        assert np.array_equal(whole[n], eng_x4.dbg_cross_attn(q[n:n + 1], kv[n:n + 1])[0])
