"""mlp_fused.hip against a float64 numpy restatement with the kernel's rounding points (bf16 LayerNorm output, bf16
hidden activation, fp32 everything else)."""
import numpy as np
import pytest
import torch
from scipy.special import erf

pytestmark = pytest.mark.gpu


def bf(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def ln(x, g, b, eps):
    x = x.astype(np.float64)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def ref_mlp(x, ln_g, ln_b, w1, b1, w2, b2, nln_g, nln_b, eps=1e-6):
    y = bf(ln(x, ln_g, ln_b, eps)).astype(np.float64)
    h = y @ bf(w1).astype(np.float64).T + b1
    h = bf(0.5 * h * (1 + erf(h / np.sqrt(2)))).astype(np.float64)
    out = x.astype(np.float64) + h @ bf(w2).astype(np.float64).T + b2
    return out, ln(out, nln_g, nln_b, eps)


@pytest.mark.parametrize("M", [128, 37, 128 * 300 + 77])
def test_mlp_fused_against_numpy(eng_bf16, M):
    rng = np.random.default_rng(M)
    x = rng.standard_normal((M, 384)).astype(np.float32) * 1.5 + rng.standard_normal((1, 384)).astype(np.float32)
    ln_g = (1 + 0.2 * rng.standard_normal(384)).astype(np.float32); ln_b = (0.1 * rng.standard_normal(384)).astype(np.float32)
    w1 = (rng.standard_normal((1536, 384)) / np.sqrt(384)).astype(np.float32); b1 = (0.2 * rng.standard_normal(1536)).astype(np.float32)
    w2 = (rng.standard_normal((384, 1536)) / np.sqrt(1536)).astype(np.float32); b2 = (0.2 * rng.standard_normal(384)).astype(np.float32)
    ng = (1 + 0.2 * rng.standard_normal(384)).astype(np.float32); nb = (0.1 * rng.standard_normal(384)).astype(np.float32)
    out, nout = eng_bf16.dbg_mlp(x, ln_g, ln_b, w1, b1, w2, b2, ng, nb)
    rows = slice(0, M) if M < 1000 else np.r_[0:300, M - 300:M]
    ro, rn = ref_mlp(x[rows], ln_g, ln_b, w1, b1, w2, b2, ng, nb)
    # bf16 boundary flips of single hidden units move an output by <= |w2| * ulp ~ 1e-3
    assert np.abs(out[rows] - ro).max() < 0.02, np.abs(out[rows] - ro).max()
    assert np.abs(nout[rows] - rn).max() < 0.04
    assert np.isfinite(out).all()
    out2, _ = eng_bf16.dbg_mlp(x, ln_g, ln_b, w1, b1, w2, b2)
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("M", [128, 61, 128 * 260 + 5])
def test_mlp_fused_with_projection_against_numpy(eng_bf16, M):
    """PROJ variant: x' = x + att . Wp^T + bp in front of the MLP, in the same launch."""
    rng = np.random.default_rng(M + 1)
    x = rng.standard_normal((M, 384)).astype(np.float32) * 1.5
    att = rng.standard_normal((M, 384)).astype(np.float32)
    wp = (rng.standard_normal((384, 384)) / np.sqrt(384)).astype(np.float32); bp = (0.2 * rng.standard_normal(384)).astype(np.float32)
    ln_g = (1 + 0.2 * rng.standard_normal(384)).astype(np.float32); ln_b = (0.1 * rng.standard_normal(384)).astype(np.float32)
    w1 = (rng.standard_normal((1536, 384)) / np.sqrt(384)).astype(np.float32); b1 = (0.2 * rng.standard_normal(1536)).astype(np.float32)
    w2 = (rng.standard_normal((384, 1536)) / np.sqrt(1536)).astype(np.float32); b2 = (0.2 * rng.standard_normal(384)).astype(np.float32)
    ng = (1 + 0.2 * rng.standard_normal(384)).astype(np.float32); nb = (0.1 * rng.standard_normal(384)).astype(np.float32)
    out, nout = eng_bf16.dbg_mlp(x, ln_g, ln_b, w1, b1, w2, b2, ng, nb, att=att, wp=wp, bp=bp)
    rows = np.r_[0:M] if M < 1000 else np.r_[0:300, M - 300:M]
    x1 = (x[rows].astype(np.float64) + bf(att[rows]).astype(np.float64) @ bf(wp).astype(np.float64).T + bp).astype(np.float32)
    ro, rn = ref_mlp(x1, ln_g, ln_b, w1, b1, w2, b2, ng, nb)
    assert np.abs(out[rows] - ro).max() < 0.02, np.abs(out[rows] - ro).max()
    assert np.abs(nout[rows] - rn).max() < 0.04
    assert np.isfinite(out).all()
    out2, _ = eng_bf16.dbg_mlp(x, ln_g, ln_b, w1, b1, w2, b2, att=att, wp=wp, bp=bp)
    assert np.array_equal(out, out2)


def test_mlp_fused_layernorm_statistics_with_a_large_row_offset(eng_bf16):
    """The front and the epilogue gather a row's LayerNorm statistics in one pass (sums shifted by the lane's first value, merged
    over the row's four lanes as mean / M2 pairs).  Rows whose mean is 200 standard deviations away from zero - where a plain
    sum-of-squares formula loses every digit in fp32 - must still normalise like the two-pass reference."""
    M = 256
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((M, 384)) * 0.5 + rng.uniform(-100.0, 100.0, (M, 1))).astype(np.float32)
    att = rng.standard_normal((M, 384)).astype(np.float32)
    wp = (rng.standard_normal((384, 384)) / np.sqrt(384)).astype(np.float32); bp = (0.2 * rng.standard_normal(384)).astype(np.float32)
    ln_g = (1 + 0.2 * rng.standard_normal(384)).astype(np.float32); ln_b = (0.1 * rng.standard_normal(384)).astype(np.float32)
    w1 = (rng.standard_normal((1536, 384)) / np.sqrt(384)).astype(np.float32); b1 = (0.2 * rng.standard_normal(1536)).astype(np.float32)
    w2 = (rng.standard_normal((384, 1536)) / np.sqrt(1536)).astype(np.float32); b2 = (0.2 * rng.standard_normal(384)).astype(np.float32)
    ng = (1 + 0.2 * rng.standard_normal(384)).astype(np.float32); nb = (0.1 * rng.standard_normal(384)).astype(np.float32)
    out, nout = eng_bf16.dbg_mlp(x, ln_g, ln_b, w1, b1, w2, b2, ng, nb, att=att, wp=wp, bp=bp)
    x1 = (x.astype(np.float64) + bf(att).astype(np.float64) @ bf(wp).astype(np.float64).T + bp).astype(np.float32)
    ro, rn = ref_mlp(x1, ln_g, ln_b, w1, b1, w2, b2, ng, nb)
    # the residual stream is ~100: one fp32 ulp of it is 8e-6, the bf16 operands' flips stay as above; the next LayerNorm divides by a std ~1
    assert np.abs(out - ro).max() < 0.03, np.abs(out - ro).max()
    assert np.abs(nout - rn).max() < 0.06, np.abs(nout - rn).max()
