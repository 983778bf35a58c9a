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


def _ref_qkv_attn_f64(x, w, b):
    N = x.shape[0]
    qkv = (x.astype(np.float64).reshape(-1, 384) @ w.astype(np.float64).T + b.astype(np.float64)).reshape(N, 128, 3, 6, 64)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    s = np.einsum("nqhd,nkhd->nhqk", q, k) * 0.125
    p = np.exp(s - s.max(-1, keepdims=True))
    o = np.einsum("nhqk,nkhd->nqhd", p, v) / p.sum(-1).transpose(0, 2, 1)[..., None]
    return o.reshape(N, 128, 384)


@pytest.mark.parametrize("N", [1, 7, 43, 300])
def test_x4_qkv_attn_one_launch_against_float64(eng_x4, N):
    """gemm_sp.hip's attention epilogue (the default precision's encoder: qkv projection + self-attention of a (crop, head) as one
    128 x 192 tile; timm Attention.forward inside the module run at /root/reference/tuatara.cpp:307): fp32-equivalent against float64 -
    1 crop = 6 tiles on 6 workgroups, 43 = a ragged last round, 300 = 1800 tiles (seven per workgroup: the persistent loop, ring
    wrap-around across tiles, the re-read of the next tile's first fragments)."""
    rng = np.random.default_rng(100 + N)
    x = rng.standard_normal((N, 128, 384)).astype(np.float32)
    if N > 1:
        x[1] *= 3.0                                                       # peaky softmax rows
    w = (rng.standard_normal((1152, 384)) / np.sqrt(384)).astype(np.float32) * 1.5
    b = (0.3 * rng.standard_normal(1152)).astype(np.float32)
    out = eng_x4.dbg_qkv_attn(x, w, b)
    out2 = eng_x4.dbg_qkv_attn(x, w, b)
    assert np.array_equal(out, out2)                                      # no race: bit-identical reruns
    ref = _ref_qkv_attn_f64(x, w, b)
    err = np.abs(out - ref)
    assert np.isfinite(out).all()
    # fp32 noise: ~1e-6 on the projections (|qkv| ~ 1.5 .. 5), ~1e-5 on scores of magnitude ~100 before the 1/8, softmax-weighted sums of |v| ~ 1.5;
    # the crop scaled by 3 has scores nine times as large (|s| up to ~1000: fp32's own rounding of a 64-term sum there is ~1e-4)
    per_crop = err.reshape(N, -1).max(1)
    tol = np.full(N, 4e-5)
    if N > 1:
        tol[1] = 6e-4
    assert (per_crop < tol).all(), per_crop.tolist()[:8]
    assert np.delete(err, 1, 0).mean() < 2e-6 if N > 1 else err.mean() < 2e-6


@pytest.mark.parametrize("N", [1, 7, 43, 300])
def test_x4_qkv_attn_four_wave_tiles_against_float64(eng_x4, N):
    """qkv_attn4.hip (tuning key qkv_attn4, off by default): the same tile on four-wave workgroups, two per CU, tiles handed out by counter - the bars of the eight-wave
    form above, bit-identical reruns, and the SAME BITS as the eight-wave kernel (same MFMAs per accumulator in the same order).  How the tiles
    are handed out and how the two workgroups of a CU take turns is not arithmetic: equal shares by stride (+ 2), no issue priority (+ 4), the per-CU
    matrix-phase token (+ 64) give the same bits."""
    rng = np.random.default_rng(100 + N)
    x = rng.standard_normal((N, 128, 384)).astype(np.float32)
    if N > 1:
        x[1] *= 3.0
    w = (rng.standard_normal((1152, 384)) / np.sqrt(384)).astype(np.float32) * 1.5
    b = (0.3 * rng.standard_normal(1152)).astype(np.float32)
    outs = {}
    try:
        for v in (0, 1, 3, 5, 65):
            assert eng_x4.set_tuning(b"qkv_attn4", v) == 0
            outs[v] = eng_x4.dbg_qkv_attn(x, w, b)
        assert eng_x4.set_tuning(b"qkv_attn4", 1) == 0
        out2 = eng_x4.dbg_qkv_attn(x, w, b)
    finally:
        eng_x4.set_tuning(b"qkv_attn4", 0)
    base, out = outs[0], outs[1]
    assert np.array_equal(out, out2)
    for v in (3, 5, 65):
        assert np.array_equal(out, outs[v]), v
    assert np.isfinite(out).all()
    ref = _ref_qkv_attn_f64(x, w, b)
    err = np.abs(out - ref)
    per_crop = err.reshape(N, -1).max(1)
    tol = np.full(N, 4e-5)
    if N > 1:
        tol[1] = 6e-4
    assert (per_crop < tol).all(), per_crop.tolist()[:8]
    assert np.delete(err, 1, 0).mean() < 2e-6 if N > 1 else err.mean() < 2e-6
    assert np.array_equal(out, base)          # every accumulator sees the eight-wave kernel's MFMAs in its order


def test_x4_encoder_on_four_wave_qkv_attention_tiles(eng_x4):
    """whole recogniser, 200 crops (tiled activation planes in and out of the launch): qkv_attn4.hip against the eight-wave tile - the same logits, bit for bit"""
    rng = np.random.default_rng(6)
    crops = rng.integers(0, 256, (200, 32, 128, 3), dtype=np.uint8)
    try:
        assert eng_x4.set_tuning(b"qkv_attn4", 0) == 0
        la, ia = eng_x4.parseq_logits(crops)
        assert eng_x4.set_tuning(b"qkv_attn4", 1) == 0
        lb, ib = eng_x4.parseq_logits(crops)
    finally:
        eng_x4.set_tuning(b"qkv_attn4", 0)
    assert np.array_equal(ia, ib)
    assert np.array_equal(la, lb)


def test_x4_encoder_with_and_without_the_fused_qkv_attention(eng_x4):
    """whole recogniser, 200 crops: the one-launch qkv + attention against the separate GEMM + attention kernels (same arithmetic up to
    the rounding of K / V to pairs: round-to-nearest in the epilogue, truncation in the stored triples) - logits within fp32 noise, ids equal"""
    rng = np.random.default_rng(5)
    crops = rng.integers(0, 256, (200, 32, 128, 3), dtype=np.uint8)
    try:
        assert eng_x4.set_tuning(b"qkv_attn_split", 0) == 0
        la, ia = eng_x4.parseq_logits(crops)
        assert eng_x4.set_tuning(b"qkv_attn_split", 1) == 0
        lb, ib = eng_x4.parseq_logits(crops)
    finally:
        eng_x4.set_tuning(b"qkv_attn_split", 1)
    assert np.array_equal(ia, ib)
    # two fp32-equivalent paths, each within 6e-4 of the fp32 oracle at |logit| ~ 32 (tests/test_gpu_x4_parity.py): measured 5.5e-4 apart at one of
    # 200 x 26 x 95 logits (the bound is the sibling comparisons' in tests/test_gpu_split_gemm.py plus that margin, not a kernel property)
    assert np.abs(la - lb).max() < 7e-4, float(np.abs(la - lb).max())
