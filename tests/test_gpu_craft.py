"""-m gpu: BASELINE.json config 3 (first half) — CRAFT heat map, engine vs CPU fp32 oracle."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines_random(weights_random):
    from tests.conftest import _engine
    return _engine(weights_random["dir"], "f32"), _engine(weights_random["dir"], "bf16")


def test_craft_small_vs_golden(engines_random):
    e32, _ = engines_random
    canvas = np.random.default_rng(1).integers(0, 256, (64, 96, 3), dtype=np.uint8)
    heat = e32.craft_heatmap(canvas)
    g = np.load(os.path.join(GOLDEN, "g6_craft.npz"))["heat"]
    assert heat.shape == g.shape == (32, 48, 2)
    assert np.abs(heat - g).max() < 1e-3, np.abs(heat - g).max()


@pytest.mark.parametrize("hw", [(256, 192), (96, 160)])
def test_craft_f32_vs_oracle_random_weights(engines_random, weights_random, hw):
    """Fully random weights: every one of the 27 convs, the pools, the upsamples and the concats
    contributes to the output, so a wrong layer cannot hide."""
    from oracle import pipeline
    e32, ebf = engines_random
    craft, _ = pipeline.load_models(weights_random["craft"], weights_random["parseq"])
    canvas = np.random.default_rng(hw[0]).integers(0, 256, (*hw, 3), dtype=np.uint8)
    ref = pipeline.craft_heatmap(craft, canvas)
    got = e32.craft_heatmap(canvas)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() < 1e-3 * max(1.0, scale), (np.abs(got - ref).max(), scale)
    gb = ebf.craft_heatmap(canvas)
    rel = np.abs(gb - ref).max() / scale
    print(f"bf16 CRAFT heat map {hw}: max|d|/max|ref| = {rel:.4f}")
    assert rel < 0.08


def test_craft_full_page_f32_and_bf16(eng_f32, eng_bf16, oracle_models, funsd):
    """1024x768 canvas of the FUNSD page, structured weights: heat map within 1e-3 (f32); bf16 reported."""
    from oracle import pipeline, post
    canvas, _ = post.resize_aspect_ratio(np.ascontiguousarray(funsd[:, :, ::-1]))
    assert canvas.shape == (1024, 768, 3)
    ref = pipeline.craft_heatmap(oracle_models[0], canvas)
    got = eng_f32.craft_heatmap(canvas)
    assert got.shape == (512, 384, 2)
    assert np.abs(got - ref).max() < 1e-3, np.abs(got - ref).max()
    gb = eng_bf16.craft_heatmap(canvas)
    print(f"bf16 CRAFT full page: max|d| = {np.abs(gb - ref).max():.4f}")
    assert np.abs(gb - ref).max() < 0.06


@pytest.mark.parametrize("hw", [(256, 192), (64, 96), (1024, 768)])
def test_craft_bf16_kernel_generations_agree(engines_random, hw):
    """bf16 mode, random weights: the second-generation kernels (conv3p / gemm2 with fused pools / conv3s head /
    conv1_direct) against the first-generation path (igemm for every conv, separate pool kernels).  Same rounding
    points (bf16 activations between layers), so the heat maps differ only by fp32 summation order and the bf16
    boundary flips that causes."""
    _, ebf = engines_random
    canvas = np.random.default_rng(hw[1]).integers(0, 256, (*hw, 3), dtype=np.uint8)
    try:
        ebf.lib.ttr_set_gemm_config(-1)
        old = ebf.craft_heatmap(canvas)
    finally:
        ebf.lib.ttr_set_gemm_config(0)
    new = ebf.craft_heatmap(canvas)
    rel = np.abs(new - old).max() / np.abs(old).max()
    print(f"bf16 CRAFT {hw}: second vs first generation kernels max|d|/max = {rel:.5f}")
    assert rel < 0.02


@pytest.mark.parametrize("hw", [(256, 192), (64, 96), (1024, 768), (32, 32)])
def test_first_pair_wave_specialised_kernel_is_bit_identical(engines_random, hw):
    """conv3p_first2s_kernel (consumer waves multiply, producer waves run conv1_1 on the next patch's halo from their own canvas
    strip; pooling before bias + ReLU) against conv3p_first2_kernel: same products in the same order, max-pool commuted with a
    monotone map — the heat maps must be equal bit for bit.  Sizes: fewer patches than CUs, a single patch row, a persistent loop
    of 12 patches per workgroup, four patches one above the other (image one patch wide)."""
    _, ebf = engines_random
    canvas = np.random.default_rng(hw[0] + hw[1]).integers(0, 256, (*hw, 3), dtype=np.uint8)
    try:
        assert ebf.set_tuning(b"c3_first_persistent", 1) == 0
        old = ebf.craft_heatmap(canvas)
        assert ebf.set_tuning(b"c3_first_persistent", 2) == 0
        new = ebf.craft_heatmap(canvas)
    finally:
        ebf.set_tuning(b"c3_first_persistent", 2)
    assert np.isfinite(new).all()
    assert np.array_equal(old, new)


@pytest.mark.parametrize("hw", [(256, 192), (32, 64), (96, 160)])
def test_upsample_block_kernel_is_bit_identical(engines_random, hw):
    """upsample2x_block_kernel (a thread forms a 2 x 4 output block from one clamped 3 x 4 input window) evaluates the expression of
    the one-output-per-thread kernel on the same source pixels: the heat maps are equal bit for bit, in both precisions, down to the
    smallest map the detector upsamples (2 x 4 at a 32 x 64 canvas: every pixel is a border pixel)."""
    canvas = np.random.default_rng(sum(hw)).integers(0, 256, (*hw, 3), dtype=np.uint8)
    for eng in engines_random:
        try:
            assert eng.set_tuning("upsample_block", 0) == 0
            a = eng.craft_heatmap(canvas)
        finally:
            eng.set_tuning("upsample_block", 1)
        b = eng.craft_heatmap(canvas)
        assert np.isfinite(a).all() and np.array_equal(a, b)
