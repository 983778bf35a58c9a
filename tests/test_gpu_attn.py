"""Encoder self-attention kernels against numpy (float64 softmax with the kernels' rounding points: bf16 inputs, P rounded to
bf16, normaliser = sum of the rounded P)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def bf(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def ref_attn(qkv):
    N = qkv.shape[0]
    x = bf(qkv).astype(np.float64).reshape(N, 128, 3, 6, 64)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]                       # [N,128,6,64]
    s = np.einsum("nqhd,nkhd->nhqk", q, k)
    p = np.exp((s - s.max(-1, keepdims=True)) * 0.125)
    p = bf(p).astype(np.float64)
    o = np.einsum("nhqk,nkhd->nqhd", p, v) / p.sum(-1).transpose(0, 2, 1)[..., None]
    return o.reshape(N, 128, 384)


@pytest.mark.parametrize("impl", [1, 0])
def test_attn_enc_bf16_against_numpy(eng_bf16, impl):
    rng = np.random.default_rng(3)
    qkv = rng.standard_normal((5, 128, 1152)).astype(np.float32) * 1.5
    qkv[1] *= 4.0                                                        # peaky softmax rows
    try:
        assert eng_bf16.set_tuning(b"attn_impl", impl) == 0
        out = eng_bf16.dbg_attn_enc(qkv)
        out2 = eng_bf16.dbg_attn_enc(qkv)
    finally:
        eng_bf16.set_tuning(b"attn_impl", 1)
    ref = ref_attn(qkv)
    assert np.array_equal(out, out2)
    assert np.isfinite(out).all()
    err = np.abs(out - ref)
    tol = 2.0 ** -7 * np.abs(ref) + 4e-3                                 # a bf16 ulp of the output + exp / summation-order noise
    assert (err <= tol).all(), float((err - tol).max())
    assert err.mean() < 2e-3


def test_attn_enc_generations_agree(eng_bf16):
    rng = np.random.default_rng(4)
    qkv = rng.standard_normal((7, 128, 1152)).astype(np.float32) * 2.0
    try:
        eng_bf16.set_tuning(b"attn_impl", 0)
        a = eng_bf16.dbg_attn_enc(qkv)
        eng_bf16.set_tuning(b"attn_impl", 1)
        b = eng_bf16.dbg_attn_enc(qkv)
    finally:
        eng_bf16.set_tuning(b"attn_impl", 1)
    assert (np.abs(a - b) <= 2.0 ** -7 * np.abs(a) + 1e-6).all()          # at most one bf16 ulp (fp32 summation order)
    assert (a != b).mean() < 0.05


@pytest.mark.parametrize("N", [3, 41, 95])
def test_qkv_attn_fused_against_numpy(eng_bf16, N):
    """qkv_attn.hip: projection + attention in one kernel; 3 crops = fewer than crop groups, 41 / 95 = ragged shares."""
    rng = np.random.default_rng(N)
    x = rng.standard_normal((N, 128, 384)).astype(np.float32)
    w = (rng.standard_normal((1152, 384)) / np.sqrt(384)).astype(np.float32) * 1.5
    b = (0.3 * rng.standard_normal(1152)).astype(np.float32)
    out = eng_bf16.dbg_qkv_attn(x, w, b)
    out2 = eng_bf16.dbg_qkv_attn(x, w, b)
    assert np.array_equal(out, out2)
    qkv = bf(x).astype(np.float64).reshape(-1, 384) @ bf(w).astype(np.float64).T + b
    ref = ref_attn(qkv.reshape(N, 128, 1152).astype(np.float32))
    err = np.abs(out - ref)
    tol = 2.0 ** -7 * np.abs(ref) + 6e-3
    assert (err <= tol).all(), float((err - tol).max())
    assert err.mean() < 2e-3
