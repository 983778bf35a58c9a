"""The split-operand linear kernels on their own (gemm_sp.hip's streamlined pairs kernels and gemm2.hip's split mode behind launch_gemm2): products of
f16 activation pairs / triples with weight pairs must reproduce the fp32 linear layer the reference runs inside its TorchScript PARSeq
(/root/reference/tuatara.cpp:307) on ragged shapes - rows and channels that do not fill the last tile, every tile configuration, every epilogue."""
import math
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


@pytest.fixture(scope="module")
def eng(tmp_path_factory):
    from tuatara_amd import weights as W
    from tuatara_amd.engine import Engine
    d = str(tmp_path_factory.mktemp("w"))
    W.make_synthetic_weights(d, seed=0, structured=True)
    e = Engine(d, precision="f16x4")
    yield e
    e.close()


def _ref(x, w, b, act, resid):
    y = x.astype(np.float64) @ w.astype(np.float64).T
    if b is not None:
        y = y + b.astype(np.float64)
    if resid is not None:
        y = y + resid.astype(np.float64)
    if act == ACT_RELU:
        y = np.maximum(y, 0.0)
    elif act == ACT_GELU:
        y = 0.5 * y * (1.0 + np.vectorize(math.erf)(y / math.sqrt(2.0)))
    return y


def _case(eng, M, K, N, npd, act, planes, with_resid, cfg, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32) if with_resid else None
    got = eng.dbg_split_gemm(x, w, b, np_products=npd, act=act, out_planes=planes, resid=r, cfg=cfg)
    ref = _ref(x, w, b, act, r)
    # fp32-equivalent: the error of an fp32 dot product of K terms of size ~1 (pairs drop the 24th bit of half the activations)
    tol = (6e-6 if npd == 3 else 3e-6) * math.sqrt(K / 384.0) + (1e-6 if planes == 2 else 0.0)
    err = np.abs(got - ref).max()
    assert np.isfinite(got).all() and err < tol * max(1.0, np.abs(ref).max() / 4.0), (M, K, N, npd, act, planes, with_resid, cfg, err)


# (M, K, N): rows / channels that leave partial tiles for 128 x 128, 128 x 256 and 256 x 128; K = 192 is the shortest the streamlined kernel takes
SHAPES = [(128 * 37 + 40, 384, 1152), (128 * 40, 384, 1536), (128 * 33 + 8, 1536, 384), (128 * 3 + 16, 192, 392), (128 * 21, 384, 384)]


@pytest.mark.parametrize("cfg", [0, 2, 3, 6])
@pytest.mark.parametrize("shape", SHAPES)
def test_pairs_linear_matches_fp32(eng, shape, cfg):
    """activation pairs (three MFMAs per product): plain fp32 output, every tile configuration (0 = the engine's own choice)"""
    M, K, N = shape
    _case(eng, M, K, N, 3, ACT_NONE, 0, False, cfg, 1)


@pytest.mark.parametrize("act,planes,resid", [(ACT_GELU, 2, False), (ACT_NONE, 3, False), (ACT_NONE, 0, True), (ACT_RELU, 2, False), (ACT_GELU, 0, True)])
@pytest.mark.parametrize("cfg", [3, 6])
def test_pairs_linear_epilogues(eng, act, planes, resid, cfg):
    """the epilogues PARSeq uses: GELU -> pairs (fc1), triples (qkv), fp32 + residual (fc2), on a ragged shape"""
    _case(eng, 128 * 19 + 72, 384, 1152 if planes else 392, 3, act, planes, resid, cfg, 2)


@pytest.mark.parametrize("cfg", [0, 2, 3, 6])
@pytest.mark.parametrize("shape", SHAPES[:3] + [(128 * 3 + 16, 192, 392), (128 * 5 + 8, 128, 392), (128 * 2, 256, 1152)])
def test_triples_linear_matches_fp32(eng, shape, cfg):
    """exact activation triples (four MFMAs per product), fp32 output + residual; K / 64 odd (192) stays on gemm2.hip's loop, K = 128 is the
    shortest the streamlined kernel takes (two k0)"""
    M, K, N = shape
    _case(eng, M, K, N, 4, ACT_NONE, 0, True, cfg, 3)


@pytest.mark.parametrize("act,planes", [(ACT_GELU, 3), (ACT_NONE, 3), (ACT_NONE, 0)])
def test_triples_linear_epilogues(eng, act, planes):
    """the decoder's epilogues on the triples kernel: GELU -> triples (ffn1), triples, fp32"""
    _case(eng, 128 * 9 + 40, 384, 1536 if act == ACT_GELU else 392, 4, act, planes, False, 3, 5)


def test_streamlined_and_general_kernels_agree(eng):
    """gemm_sp.hip against gemm2.hip's own K loop on the same operands: the same products in a different order"""
    rng = np.random.default_rng(4)
    M, K, N = 128 * 24, 384, 1152
    x = rng.standard_normal((M, K)).astype(np.float32); w = (rng.standard_normal((N, K)) / 20).astype(np.float32); b = rng.standard_normal(N).astype(np.float32)
    a = eng.dbg_split_gemm(x, w, b, np_products=3, cfg=6)
    assert eng.set_tuning("g2_split_stream", 0) == 0
    try:
        c = eng.dbg_split_gemm(x, w, b, np_products=3, cfg=6)
    finally:
        assert eng.set_tuning("g2_split_stream", 2) == 0
    assert np.abs(a - c).max() < 2e-6 * max(1.0, np.abs(a).max())


# (M, K, N, act, out_planes, resid): the six linears of an AR step at one page's 40 crops, and the corners of the skinny kernel
SKINNY = [(40, 384, 768, ACT_NONE, 0, False), (40, 384, 384, ACT_NONE, 0, True), (40, 384, 1536, ACT_GELU, 3, False), (40, 1536, 384, ACT_NONE, 0, True),
          (1, 384, 384, ACT_NONE, 3, False), (64, 384, 96, ACT_RELU, 2, False), (17, 128, 32, ACT_NONE, 0, False), (33, 192, 416, ACT_NONE, 0, True),
          (48, 64, 64, ACT_NONE, 0, False), (40, 384, 95, ACT_NONE, 0, False), (33, 96, 94, ACT_NONE, 0, True),   # the head: 95 classes, rows of 95 floats
          (1280, 384, 768, ACT_NONE, 0, False), (1280 + 37, 1536, 384, ACT_NONE, 0, True), (200, 384, 1536, ACT_GELU, 3, False),   # row blocks: the AR steps of a 32-page batch
          (257, 384, 96, ACT_NONE, 2, False)]


@pytest.mark.parametrize("case", SKINNY)
def test_skinny_whole_k_linear_matches_fp32(eng, case):
    """gemm_skx.hip (cfg 7): <= 64 rows, 32 output channels and the whole K per workgroup, K split over the four waves and joined in LDS -
    the linears of the AR steps of a single page (the 26 sequential decoder steps inside the module run at /root/reference/tuatara.cpp:307).
    1, 2, 3 and 4 row blocks and grids of several 64-row blocks (a ragged last one); K of 2 .. 48 steps (fewer steps than waves: K = 64); a channel
    count that leaves half a workgroup empty."""
    M, K, N, act, planes, with_resid = case
    _case(eng, M, K, N, 4, act, planes, with_resid, 7, 11)


def test_skinny_and_ring_kernels_agree_and_the_engine_picks_the_skinny_one(eng):
    rng = np.random.default_rng(12)
    M, K, N = 40, 384, 768
    x = rng.standard_normal((M, K)).astype(np.float32); w = (rng.standard_normal((N, K)) / 20).astype(np.float32); b = rng.standard_normal(N).astype(np.float32)
    a = eng.dbg_split_gemm(x, w, b, np_products=4, cfg=7)
    c = eng.dbg_split_gemm(x, w, b, np_products=4, cfg=3)
    assert np.abs(a - c).max() < 2e-6 * max(1.0, np.abs(a).max())
    # whole recogniser at one page's crop count: the AR steps on the skinny kernel against the ring kernels - same ids, logits within fp32 noise
    crops = rng.integers(0, 256, (40, 32, 128, 3), dtype=np.uint8)
    try:
        assert eng.set_tuning("skinny_split", 0) == 0
        l0, i0 = eng.parseq_logits(crops)
        assert eng.set_tuning("skinny_split", 1) == 0
        l1, i1 = eng.parseq_logits(crops)
    finally:
        eng.set_tuning("skinny_split", 1)
    assert np.array_equal(i0, i1) and np.abs(l0 - l1).max() < 6e-4      # (two fp32-equivalent paths: each within ~5e-4 of the oracle at |logit| ~ 32)


def test_skinny_layernorm_prologue_changes_nothing(eng):
    """The decoder's LayerNorm + linear pairs as ONE skinny launch (gemm_skx.hip, LayerNorm prologue; tuning key "skx_ln_fuse") against the
    LayerNorm kernel followed by the skinny linear: the same arithmetic in the same order, so the AR logits, the refined logits and the ids
    are identical bit for bit - at a page's crop count (three 16-row blocks, the last one ragged), for a single crop, and for 17."""
    _fold_check(eng, "skx_ln_fuse", (40, 1, 17, 300))      # (300 crops: the prologue form is offered up to 2048 rows)


def test_argmax_inside_the_next_steps_embedding_kernel_changes_nothing(eng):
    """An AR step's argmax (token, first-EOS count) found by the next step's embedding + LayerNorm kernel (tuning key "argmax_fold") instead of by
    its own launch: same ids, same logits, bit for bit - a page's crops (where the host also looks at the done counter every fourth step
    and the argmax keeps its own launch there), a single crop, and 300 crops (no host checks: every step folded)."""
    _fold_check(eng, "argmax_fold", (40, 1, 300))


def test_embedding_inside_the_self_kv_linear_changes_nothing(eng):
    """An AR step's token -> embedding -> norm_c -> self_kv projection as one skinny launch (tuning key "embed_fold"; gemm_skx.hip's token prologue,
    which also takes the pending argmax) against dec_embed_ln_planes_kernel + the skinny linear: identical bit for bit, with and without the
    folded argmax."""
    _fold_check(eng, "embed_fold", (40, 1, 17))
    try:
        assert eng.set_tuning("argmax_fold", 0) == 0
        _fold_check(eng, "embed_fold", (40,))
    finally:
        eng.set_tuning("argmax_fold", 1)


def test_tiled_planes_change_nothing(eng):
    """The encoder's activation planes and the recogniser's weight planes laid out as gemm_sp.hip's loader pieces (tuning keys "sp_tiled_x",
    "sp_tiled_w": 1-KiB blocks of 8 rows x 64 halves instead of row-major rows) against the row-major tensors: a layout, not an arithmetic -
    logits and ids identical bit for bit, at 1 (a group small enough for the skinny projection keeps its rows), 17, 40 and 300 crops."""
    _fold_check(eng, "sp_tiled_x", (40, 1, 17, 300))
    _fold_check(eng, "sp_tiled_w", (40, 300))


def test_hidden_planes_in_lane_order_change_nothing(eng):
    """The MLP's hidden activation (fc1 -> fc2) as 16-row pieces in the producing epilogue's lane order, one contiguous KiB per store instruction, with
    streaming stores (tuning key "sp_hidden16" = 2) and without (1), against the loader's 8-row pieces (0, the default: measured faster): a layout and a cache policy -
    logits and ids identical bit for bit at a page's crops (64-row tiles), 17, 300 (256 x 128 / 128 x 128 tiles) and 640 crops (fc1 on 128 x 256 tiles);
    also with the general epilogue kernel writing the layout ("gsp_epi" = 0)."""
    _fold_check(eng, "sp_hidden16", (40, 17, 300, 640), on=1, restore=0)
    _fold_check(eng, "sp_hidden16", (40, 300), on=2, restore=0)
    try:
        assert eng.set_tuning("gsp_epi", 0) == 0
        _fold_check(eng, "sp_hidden16", (40, 300), on=2, restore=0)
    finally:
        eng.set_tuning("gsp_epi", 3)


def test_fixed_epilogue_kernels_change_nothing(eng):
    """gemm_sp.hip's recurring epilogue cases on kernels of their own (tuning key "gsp_epi": fc1's GELU + tiled pairs, the residual linears' bias +
    residual + fp32 rows, as compile-time cases of the same code) against the general kernel that tests ConvParams' flags per 8-value block: the same
    arithmetic in the same order - logits and ids identical bit for bit, at a page's crops (64-row and 128-row tiles), 17, 300 (256 x 128 tiles) and
    640 crops (fc1 on 128 x 256 tiles)."""
    _fold_check(eng, "gsp_epi", (40, 17, 300, 640), on=15, restore=3)


def test_whole_line_stores_change_nothing(eng):
    """The encoder GEMMs' epilogues with every store instruction writing whole 128-byte lines (tuning key "gsp_epi" bits 16: fc1's tiled pairs as contiguous
    KiB pieces, 32: the residual linears' fp32 rows; lanes fr and fr ^ 8 exchange one 16-byte piece by DPP, gemm_sp.hip EM = 3 / 4) against the accumulator-layout
    stores (sixteen half lines per instruction; profiles/r06_store_rate.txt): pure data movement - logits and ids identical bit for bit at 17, 300 and 640 crops
    (128 x 128, 256 x 128 and 128 x 256 tiles) and at a page's 40 (64-row tiles: unchanged kernels)."""
    _fold_check(eng, "gsp_epi", (40, 17, 300, 640), on=51, restore=3, off=3)
    _fold_check(eng, "gsp_epi", (300,), on=3 + 16, restore=3, off=3)
    _fold_check(eng, "gsp_epi", (300,), on=3 + 32, restore=3, off=3)


def test_staggered_wave_halves_change_nothing(eng):
    """The eight-wave pairs tiles with waves 4 - 7 half a phase behind waves 0 - 3 (gemm_sp_kernel's STAG, tuning key "gsp_stag"): the same requests in the same
    barrier intervals, the same accumulation order per accumulator - logits and ids identical bit for bit at 300 and 640 crops.  (Measured slower: off by default.)"""
    _fold_check(eng, "gsp_stag", (300, 640), on=1, restore=0, off=0)


def test_ninety_six_row_tiles_change_nothing(eng):
    """A page of 43 - 128 crops: the encoder's proj / fc2 (129 - 384 tiles of 128 x 128 on 256 CUs) take 96-row tiles where those need no extra round
    (tuning key "gsp_few": 1 = with them, 2 = without): the same k order per output - logits and ids identical bit for bit at 43, 52, 60, 64 crops (one round),
    100 and 127 (two), and at 70 and 160, where the rule does not fire."""
    _fold_check(eng, "gsp_few", (43, 52, 60, 64, 70, 100, 127, 160), on=1, restore=1, off=2)


def _fold_check(eng, key, counts, on=1, restore=None, off=0):
    rng = np.random.default_rng(77)
    for n in counts:
        crops = rng.integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
        if n >= 8:
            crops[: n // 2, :, 40:] = 255          # short words too: half the crops are blank behind a third of their width
        try:
            assert eng.set_tuning(key, off) == 0
            l0, a0, i0 = eng.parseq_logits(crops, want_ar=True)
            assert eng.set_tuning(key, on) == 0
            l1, a1, i1 = eng.parseq_logits(crops, want_ar=True)
        finally:
            eng.set_tuning(key, on if restore is None else restore)
        assert np.isfinite(l1).all() and np.array_equal(i0, i1) and np.array_equal(l0, l1), n
        # AR logits: up to each crop's EOS step (behind it the attention kernels skip the crop and its rows hold whatever the buffers held)
        for c in range(n):
            eos = np.flatnonzero(i0[c] == 0)
            upto = int(eos[0]) + 1 if len(eos) else 26
            assert np.array_equal(a0[c, :upto], a1[c, :upto]), (n, c)
