"""-m gpu: op-level parity of the implicit-GEMM kernel (through the C ABI) against fp32 torch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_conv(x0, w, b, ks, dil, act, x1=None, relu0=False, relu1=False):
    xs = [torch.from_numpy(x0).permute(0, 3, 1, 2)]
    if relu0:
        xs[0] = xs[0].relu()
    if x1 is not None:
        t = torch.from_numpy(x1).permute(0, 3, 1, 2)
        xs.append(t.relu() if relu1 else t)
    x = torch.cat(xs, 1)
    wt = torch.from_numpy(w).permute(0, 3, 1, 2).contiguous()  # [Cout,kh,kw,Cin] -> OIHW
    y = F.conv2d(x, wt, torch.from_numpy(b) if b is not None else None, padding=dil * (ks // 2), dilation=dil)
    if act == 1:
        y = y.relu()
    elif act == 2:
        y = F.gelu(y)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


CASES = [
    # B, H, W, C0, C1, Cout, ks, dil, act, relu0
    (1, 8, 16, 32, 0, 64, 3, 1, 1, False),
    (2, 12, 20, 64, 0, 128, 3, 1, 0, True),     # ragged M (480), batch, relu-on-load
    (1, 16, 16, 64, 0, 96, 3, 6, 0, False),     # dilation 6 (slice5.1), Cout not a tile multiple
    (1, 10, 12, 64, 32, 64, 1, 1, 1, False),    # virtual concat 1x1 (up-blocks)
    (1, 1, 300, 96, 0, 384, 1, 1, 0, False),    # linear: patch embed shape
    (1, 1, 70, 384, 0, 95, 1, 1, 2, False),     # skinny GEMM + GELU, Cout 95
    (1, 24, 24, 32, 0, 2, 1, 1, 0, False),      # head: Cout 2
    (1, 24, 24, 32, 0, 32, 3, 1, 1, False),     # head 3x3 narrow
]


@pytest.mark.parametrize("case", CASES)
def test_conv_f32_exact(eng_f32, case):
    B, H, W, C0, C1, Cout, ks, dil, act, relu0 = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x0 = rng.standard_normal((B, H, W, C0)).astype(np.float32)
    x1 = rng.standard_normal((B, H, W, C1)).astype(np.float32) if C1 else None
    w = (rng.standard_normal((Cout, ks, ks, C0 + C1)) / np.sqrt(ks * ks * (C0 + C1))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = eng_f32.dbg_conv(x0, w, b, ks, dil, act, x1=x1, relu0=relu0)
    ref = _ref_conv(x0, w, b, ks, dil, act, x1, relu0)
    assert np.abs(got - ref).max() < 2e-5


@pytest.mark.parametrize("case", CASES)
def test_conv_bf16(eng_bf16, case):
    B, H, W, C0, C1, Cout, ks, dil, act, relu0 = case
    rng = np.random.default_rng(hash(case) % 2**31)
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    x0 = bf(rng.standard_normal((B, H, W, C0)).astype(np.float32))
    x1 = bf(rng.standard_normal((B, H, W, C1)).astype(np.float32)) if C1 else None
    w = bf((rng.standard_normal((Cout, ks, ks, C0 + C1)) / np.sqrt(ks * ks * (C0 + C1))).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    got = eng_bf16.dbg_conv(x0, w, b, ks, dil, act, x1=x1, relu0=relu0)
    ref = _ref_conv(x0, w, b, ks, dil, act, x1, relu0)   # inputs are exactly representable in bf16: only accumulation order differs
    assert np.abs(got - ref).max() < 2e-4


def test_conv_identity_asymmetric(eng_f32):
    """A = I check with an asymmetric B (catches a transposed C write)."""
    K = 64
    x = np.eye(K, dtype=np.float32).reshape(1, 1, K, K)
    w = np.arange(K * K, dtype=np.float32).reshape(K, 1, 1, K) / 100.0
    got = eng_f32.dbg_conv(x, w, None, 1)
    assert np.array_equal(got[0, 0], w.reshape(K, K).T)


# ---- gemm2.hip (LDS-DMA staged bf16 kernel): every tile configuration, same reference
G2_CASES = [
    # B, H, W, C0, C1, Cout, ks, dil, act
    (1, 20, 24, 64, 0, 64, 3, 1, 1),
    (2, 13, 17, 128, 0, 136, 3, 1, 0),      # ragged M, ragged Cout, halos at the batch boundary
    (1, 16, 20, 64, 0, 256, 3, 6, 0),       # dilation 6 (slice5.1)
    (1, 9, 31, 128, 64, 128, 1, 1, 1),      # virtual concat 1x1 (up-blocks)
    (1, 12, 12, 64, 64, 72, 3, 1, 2),       # concat + 3x3 + GELU
    (1, 1, 700, 384, 0, 1152, 1, 1, 0),     # PARSeq qkv shape
    (1, 1, 300, 1536, 0, 384, 1, 1, 2),     # PARSeq fc2 shape (+GELU)
    (1, 16, 64, 64, 0, 64, 3, 1, 1),        # conv3p-eligible (H % 8, W % 32): image-border halos on every side
    (2, 8, 32, 128, 0, 136, 3, 1, 0),       # one patch per image, two channel chunks, ragged Cout
    (1, 24, 96, 64, 0, 256, 3, 1, 1),       # 3 x 3 patches, BN 256
    (1, 16, 48, 64, 0, 64, 3, 1, 1),        # width not a multiple of 32: conv3p's 16 x 16 patches (CRAFT's 64 x 48 level)
    (2, 32, 16, 128, 0, 136, 3, 1, 0),      # 16 x 16 patches, two images, two channel chunks, ragged Cout
]


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("case", G2_CASES)
def test_gemm2_configs(eng_bf16, case, cfg):
    B, H, W, C0, C1, Cout, ks, dil, act = case
    rng = np.random.default_rng(hash(case) % 2**31)
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    x0 = bf(rng.standard_normal((B, H, W, C0)).astype(np.float32))
    x1 = bf(rng.standard_normal((B, H, W, C1)).astype(np.float32)) if C1 else None
    w = bf((rng.standard_normal((Cout, ks, ks, C0 + C1)) / np.sqrt(ks * ks * (C0 + C1))).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = _ref_conv(x0, w, b, ks, dil, act, x1)
    try:
        eng_bf16.lib.ttr_set_gemm_config(cfg)
        got = eng_bf16.dbg_conv(x0, w, b, ks, dil, act, x1=x1)
    finally:
        eng_bf16.lib.ttr_set_gemm_config(0)
    assert np.abs(got - ref).max() < 2e-4


def test_gemm2_matches_first_generation_kernel(eng_bf16):
    """Same bf16 inputs, fp32 accumulation in both kernels: results agree to summation-order noise."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 40, 36, 128)).astype(np.float32)
    w = (rng.standard_normal((256, 3, 3, 128)) / 34.0).astype(np.float32)
    b = rng.standard_normal(256).astype(np.float32)
    try:
        eng_bf16.lib.ttr_set_gemm_config(-1)
        old = eng_bf16.dbg_conv(x, w, b, 3, 1, 1)
        eng_bf16.lib.ttr_set_gemm_config(0)
        new = eng_bf16.dbg_conv(x, w, b, 3, 1, 1)
    finally:
        eng_bf16.lib.ttr_set_gemm_config(0)
    assert np.abs(old - new).max() < 1e-4


@pytest.mark.parametrize("cfg", [0, 1, 3, 5, 7])
@pytest.mark.parametrize("case", [(1, 12, 40, 64, 64, 1, False), (2, 6, 10, 128, 136, 0, True), (1, 18, 14, 64, 256, 1, False),
                                  (1, 16, 64, 64, 64, 1, False), (2, 8, 32, 128, 136, 0, True), (1, 24, 96, 64, 256, 1, False),
                                  (1, 16, 48, 64, 64, 1, False), (2, 32, 16, 128, 136, 0, True)])
def test_gemm2_fused_maxpool(eng_bf16, case, cfg):
    """CRAFT's trunk pools: the 2x2 max-pool fused into the conv epilogue equals pooling the conv's own bf16 output
    bit for bit (max commutes with rounding), and that output matches the fp32 reference."""
    B, H, W, C0, Cout, act, pool_relu = case
    rng = np.random.default_rng(hash(case) % 2**31)
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    x = bf(rng.standard_normal((B, H, W, C0)).astype(np.float32))
    w = bf((rng.standard_normal((Cout, 3, 3, C0)) / np.sqrt(9 * C0)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    try:
        eng_bf16.lib.ttr_set_gemm_config(cfg)
        full, pool = eng_bf16.dbg_conv_pool(x, w, b, 3, act, pool_relu)
        _, pool_only = eng_bf16.dbg_conv_pool(x, w, b, 3, act, pool_relu, want_full=False)
    finally:
        eng_bf16.lib.ttr_set_gemm_config(0)
    ref = _ref_conv(x, w, b, 3, 1, act)
    assert np.abs(full - ref).max() < 0.03                 # bf16 output rounding
    src = np.maximum(full, 0) if pool_relu else full
    want = src.reshape(B, H // 2, 2, W // 2, 2, Cout).max(axis=(2, 4))
    assert np.array_equal(pool, want)
    assert np.array_equal(pool_only, want)


SK_CASES = [
    # M, K, Cout, act   (B=1, H=1, W=M linear layers: the decoder's per-step GEMMs)
    (614, 384, 768, 0), (40, 384, 384, 0), (333, 384, 1536, 2), (333, 1536, 384, 0), (70, 384, 95, 0), (1, 128, 4, 1),
]


@pytest.mark.parametrize("case", SK_CASES)
def test_gemm_skinny(eng_bf16, case):
    """gemm_sk.hip (whole-K-resident skinny GEMM) against fp32 torch, and against gemm2 on the same inputs."""
    M, K, Cout, act = case
    rng = np.random.default_rng(hash(case) % 2**31)
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    x = bf(rng.standard_normal((1, 1, M, K)).astype(np.float32))
    w = bf((rng.standard_normal((Cout, 1, 1, K)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = _ref_conv(x, w, b, 1, 1, act)
    try:
        assert eng_bf16.set_tuning(b"sk_max_rows", 2048) == 0
        got = eng_bf16.dbg_conv(x, w, b, 1, 1, act)
        eng_bf16.set_tuning(b"sk_max_rows", 0)
        other = eng_bf16.dbg_conv(x, w, b, 1, 1, act)
    finally:
        eng_bf16.set_tuning(b"sk_max_rows", 2048)
    assert np.abs(got - ref).max() < 2e-4
    assert np.abs(got - other).max() < 1e-4


WS_CASES = [
    # M, K, Cout, act   (linear layers, K <= 384: the ViT qkv / proj / fc1 and the batched decoder linears)
    (700, 384, 1152, 0), (300, 384, 384, 0), (1500, 384, 1536, 2), (130, 384, 768, 0), (515, 128, 72, 1), (4100, 320, 264, 0),
]


@pytest.mark.parametrize("case", WS_CASES)
def test_gemm_weight_stationary(eng_bf16, case):
    """gemm_ws.hip (weights resident in registers, activations streamed) against fp32 torch and against gemm2."""
    M, K, Cout, act = case
    rng = np.random.default_rng(hash(case) % 2**31)
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    x = bf(rng.standard_normal((1, 1, M, K)).astype(np.float32))
    w = bf((rng.standard_normal((Cout, 1, 1, K)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = _ref_conv(x, w, b, 1, 1, act)
    try:
        eng_bf16.set_tuning(b"sk_max_rows", 0)
        eng_bf16.set_tuning(b"ws_min_rows", 1)
        got = eng_bf16.dbg_conv(x, w, b, 1, 1, act)
        eng_bf16.set_tuning(b"ws_min_rows", 0)
        other = eng_bf16.dbg_conv(x, w, b, 1, 1, act)
    finally:
        eng_bf16.set_tuning(b"sk_max_rows", 2048)
        eng_bf16.set_tuning(b"ws_min_rows", 8192)
    assert np.abs(got - ref).max() < 2e-4
    assert np.abs(got - other).max() < 1e-4


@pytest.mark.parametrize("case", [(40 * 64 * 7 + 29, 384, 1536, 2), (20000, 384, 1152, 0), (9000, 256, 520, 1)])
def test_gemm_weight_stationary_bf16_output_long_streams(eng_bf16, case):
    """The engine's use of gemm_ws: bf16 output through counted buffer stores (the activation-panel wait counts past the
    epilogue's stores), every workgroup streaming several panels, ragged last panel and ragged last column slice.
    bf16 results must equal gemm2's up to fp32 summation order (<= 1 bf16 ulp on a few elements)."""
    M, K, Cout, act = case
    rng = np.random.default_rng(5)
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    x = bf(rng.standard_normal((1, 1, M, K)).astype(np.float32))
    w = bf((rng.standard_normal((Cout, 1, 1, K)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    try:
        eng_bf16.set_tuning(b"dbg_bf16_out", 1)
        eng_bf16.set_tuning(b"ws_min_rows", 1)
        got = eng_bf16.dbg_conv(x, w, b, 1, 1, act)
        again = eng_bf16.dbg_conv(x, w, b, 1, 1, act)
        eng_bf16.set_tuning(b"ws_min_rows", 0)
        other = eng_bf16.dbg_conv(x, w, b, 1, 1, act)
    finally:
        eng_bf16.set_tuning(b"dbg_bf16_out", 0)
        eng_bf16.set_tuning(b"ws_min_rows", 8192)
    assert np.array_equal(got, again)                                   # no race: run to run identical
    scale = np.maximum(np.abs(other), 1.0)
    assert (np.abs(got - other) / scale).max() < 2 ** -6                # one bf16 ulp
    assert (got != other).mean() < 0.02
    ref = _ref_conv(x[:, :, :512], w, b, 1, 1, act)
    assert np.abs(got[:, :, :512] - ref).max() < 0.04
