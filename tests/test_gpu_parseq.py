"""-m gpu: BASELINE.json config 2 — PARSeq-only, seeded random crops, engine vs CPU fp32 oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same_path(a0, a1):
    """Per crop: the AR token paths of two runs agree up to and including the first run's EOS (steps behind a batch's early exit
    read zero in both, and a fork can move the exit step)."""
    from tests.parity_rules import upto_eos
    t0, t1 = a0.argmax(-1), a1.argmax(-1)
    mask = np.arange(t0.shape[1])[None, :] < upto_eos(t0)[:, None]
    return ((t0 == t1) | ~mask).all(1)


def _ar_diff(a0, a1, same):
    """|a1 - a0| of the crops in `same` over the AR steps that ran in both runs, per crop up to and including its EOS: behind a crop's
    EOS the step's attention kernels skip it (its rows hold whatever the previous step left), and nothing reads those logits."""
    from tests.parity_rules import upto_eos
    ran = (np.abs(a0).max((0, 2)) > 0) & (np.abs(a1).max((0, 2)) > 0)
    live = np.arange(a0.shape[1])[None, :] < upto_eos(a0.argmax(-1))[:, None]
    d = np.abs(a1 - a0) * live[:, :, None]
    return d[same][:, ran]


def _crops(n, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)


def test_parseq_f32_logits_within_1e3(eng_f32, oracle_models):
    """north_star tolerance: logits within 1e-3 of the CPU reference path (parity mode = fp32 MFMA)."""
    from oracle import pipeline, post
    _, parseq = oracle_models
    crops = _crops(24)
    import torch
    with torch.no_grad():
        x = torch.from_numpy(crops).permute(0, 3, 1, 2).float().div(255.0)
        ref, ref_ar = parseq(x, return_ar=True)
    ref, ref_ar = ref.numpy(), ref_ar.numpy()
    got, got_ar, ids = eng_f32.parseq_logits(crops, want_ar=True)
    # AR logits are defined per crop up to and including its EOS step (the engine follows upstream's early exit from the AR loop)
    from tests.parity_rules import upto_eos
    live = np.arange(26)[None, :] < upto_eos(ref_ar.argmax(-1))[:, None]
    assert np.abs(got_ar - ref_ar)[live].max() < 1e-3, np.abs(got_ar - ref_ar)[live].max()
    assert np.abs(got - ref).max() < 1e-3, np.abs(got - ref).max()
    s_ref, ids_ref = post.decode_logits(ref)
    assert np.array_equal(ids, ids_ref)
    from tuatara_amd.engine import decode_ids
    assert [decode_ids(r) for r in ids] == s_ref


def test_parseq_f32_batch_invariance(eng_f32):
    crops = _crops(9, seed=3)
    a, _ = eng_f32.parseq_logits(crops)
    b = np.concatenate([eng_f32.parseq_logits(crops[i:i + 1])[0] for i in range(9)])
    assert np.abs(a - b).max() < 1e-4


def test_parseq_bf16_margin_rule_small_batch(eng_bf16, oracle_models):
    """bf16 throughput mode, small-batch kernels (48 crops): the margin rule of tests/parity_rules.py against the CPU oracle.
    (The benchmark's batch sizes: tests/test_gpu_bf16_parity.py.)"""
    import torch
    from tests import parity_rules as R
    _, parseq = oracle_models
    crops = _crops(48, seed=1)
    with torch.no_grad():
        x = torch.from_numpy(crops).permute(0, 3, 1, 2).float().div(255.0)
        ref, ref_ar = parseq(x, return_ar=True)
    ref, ref_ar = ref.numpy(), ref_ar.numpy()
    got, got_ar, ids = eng_bf16.parseq_logits(crops, want_ar=True)
    st = R.parseq_margin_rule(ref, ref_ar, got, got_ar, min_same=0.85, label="bf16 PARSeq, 48 crops vs oracle")
    assert st["mean_dlogit"] < 0.5


@pytest.mark.parametrize("G", [4, 8, 16])
def test_fused_ar_decoder_matches_kernel_per_op_loop(eng_bf16, G):
    """dec_fused.hip (one persistent kernel for the 26-step AR loop) vs the kernel-per-op schedule, both bf16:
    same greedy tokens, same refined logits; AR logits differ only by fp32 summation order / bf16 boundary flips."""
    rng = np.random.default_rng(11)
    crops = rng.integers(0, 256, (37, 32, 128, 3), dtype=np.uint8)     # 37: ragged last workgroup for every G
    try:
        eng_bf16.set_tuning(b"ar_early_exit", 0)            # the fused kernel always runs the 26 steps: compare like with like
        eng_bf16.set_tuning(b"decoder_mode", 0)
        l0, a0, i0 = eng_bf16.parseq_logits(crops, want_ar=True)
        eng_bf16.set_tuning(b"decoder_mode", G)
        l1, a1, i1 = eng_bf16.parseq_logits(crops, want_ar=True)
    finally:
        eng_bf16.set_tuning(b"decoder_mode", 1)
        eng_bf16.set_tuning(b"ar_early_exit", 1)
    assert np.isfinite(l1).all() and np.isfinite(a1).all()
    d0 = np.abs(a1[:, 0] - a0[:, 0]).max(1)                            # step 0: no token feedback yet
    assert np.median(d0) < 1e-3 and d0.max() < 0.05                    # fp32 summation order; a crop may catch one bf16 boundary flip
    same_path = (a0.argmax(-1) == a1.argmax(-1)).all(1)
    assert same_path.mean() >= 0.9                                     # greedy paths may fork only at near-ties
    # (the maxima sit on crops with a content bit inside its soft knee, where a bf16 boundary flip upstream moves every logit)
    assert np.percentile(np.abs(a1[same_path] - a0[same_path]), 99.9) < 0.25 and np.abs(a1[same_path] - a0[same_path]).max() < 1.5
    assert np.percentile(np.abs(l1[same_path] - l0[same_path]), 99.9) < 0.25 and np.abs(l1[same_path] - l0[same_path]).max() < 1.5
    from tests.parity_rules import upto_eos
    keep = np.arange(26)[None, :] < upto_eos(i0)[:, None]
    assert np.array_equal(i0[same_path][keep[same_path]], i1[same_path][keep[same_path]])


def test_layernorm_fused_into_skinny_gemm_matches_separate_kernels(eng_bf16):
    """gemm_sk's LayerNorm prologue (decoder norm1 / norm2 / decoder.norm at AR steps) vs layernorm_kernel + GEMM:
    same bf16 rounding point, only the fp32 reduction order of mean / variance differs."""
    rng = np.random.default_rng(12)
    crops = rng.integers(0, 256, (45, 32, 128, 3), dtype=np.uint8)     # 45 rows: ragged 32-row tile
    try:
        assert eng_bf16.set_tuning(b"ln_fuse", 0) == 0
        l0, a0, i0 = eng_bf16.parseq_logits(crops, want_ar=True)
        assert eng_bf16.set_tuning(b"ln_fuse", 1) == 0
        l1, a1, i1 = eng_bf16.parseq_logits(crops, want_ar=True)
    finally:
        eng_bf16.set_tuning(b"ln_fuse", 1)
    assert np.isfinite(a1).all()
    assert np.abs(a1[:, 0] - a0[:, 0]).max() < 0.05                    # step 0: no token feedback yet; bf16 boundary flips only
    same_path = _same_path(a0, a1)
    assert same_path.mean() >= 0.9
    assert np.percentile(_ar_diff(a0, a1, same_path), 99.9) < 0.25 and _ar_diff(a0, a1, same_path).max() < 1.5
    assert np.array_equal(i0[same_path], i1[same_path])


def test_refinement_self_attention_per_crop_kernel_is_bit_identical(eng_bf16):
    """dec_self_attn_refine_kernel (one workgroup per crop, K/V cache in LDS, all 26 query rows) against dec_self_attn_kernel (one
    workgroup per row): same operations in the same order — logits and ids must be equal bit for bit.  The crops' random tokens
    include early EOS for some (key padding) and none for others."""
    rng = np.random.default_rng(15)
    crops = rng.integers(0, 256, (45, 32, 128, 3), dtype=np.uint8)
    try:
        assert eng_bf16.set_tuning(b"self_refine", 0) == 0
        l0, i0 = eng_bf16.parseq_logits(crops)
        assert eng_bf16.set_tuning(b"self_refine", 1) == 0
        l1, i1 = eng_bf16.parseq_logits(crops)
    finally:
        eng_bf16.set_tuning(b"self_refine", 1)
    assert np.isfinite(l1).all()
    assert np.array_equal(l0, l1) and np.array_equal(i0, i1)


@pytest.mark.parametrize("n", [45, 700])
def test_refinement_block_through_fused_kernel_matches_separate_kernels(eng_bf16, n):
    """Refinement pass: cross_out + residual + norm2 + linear1 + GELU + linear2 + residual + final norm through mlp_fused.hip (the
    encoder block kernel with the decoder's weights; 26 n rows, a ragged last panel) against the separate GEMM / LayerNorm
    kernels.  The AR pass is untouched, so both runs refine the same token sequences: same rounding points, fp32 summation order
    differs.  700 crops = above the row count from which the engine picks the fused kernel by itself."""
    rng = np.random.default_rng(16)
    crops = rng.integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
    try:
        assert eng_bf16.set_tuning(b"dec_mlp_fused", 0) == 0
        l0, a0, i0 = eng_bf16.parseq_logits(crops, want_ar=True)
        assert eng_bf16.set_tuning(b"dec_mlp_fused", 2) == 0            # 2: the fused block whatever the panel count (1 leaves it to the engine, which
        assert eng_bf16.set_tuning(b"dec_mlp_min_rows", 1) == 0         # declines when the panels fill the last round of CUs badly)
        l1, a1, i1 = eng_bf16.parseq_logits(crops, want_ar=True)
    finally:
        eng_bf16.set_tuning(b"dec_mlp_fused", 1)
        eng_bf16.set_tuning(b"dec_mlp_min_rows", 16384)
    assert np.isfinite(l1).all()
    assert np.array_equal(a0, a1)                                      # the AR pass does not use the fused block
    d = np.abs(l1 - l0)
    print(f"refinement block fused vs separate ({n} crops): median |dlogit| {np.median(d):.4f}, max {d.max():.3f}, logit sigma {l0.std():.2f}")
    assert np.median(d) < 0.01 and d.max() < 0.1                       # measured: median 0, max 0.02 at logit sigma 6.5
    assert (i0 == i1).mean() > 0.97


@pytest.mark.parametrize("n", [45, 3])
def test_refinement_cross_attention_on_matrix_cores_matches_per_row_kernel(eng_bf16, n):
    """dec_cross_attn_mfma_kernel (attn_dec2.hip: one wave per crop and head pair, S^T = K Q^T on the matrix cores, P rounded to bf16
    as in the encoder attention) against dec_cross_attn_rows_kernel (one workgroup per query row, P in fp32).  The AR pass is
    untouched, so both runs refine the same token sequences."""
    rng = np.random.default_rng(17)
    crops = rng.integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
    try:
        assert eng_bf16.set_tuning(b"cross_mfma", 0) == 0
        l0, a0, i0 = eng_bf16.parseq_logits(crops, want_ar=True)
        assert eng_bf16.set_tuning(b"cross_mfma", 1) == 0
        l1, a1, i1 = eng_bf16.parseq_logits(crops, want_ar=True)
    finally:
        eng_bf16.set_tuning(b"cross_mfma", 1)
    assert np.isfinite(l1).all()
    assert np.array_equal(a0, a1)
    d = np.abs(l1 - l0)
    print(f"refinement cross-attention MFMA vs per-row ({n} crops): median |dlogit| {np.median(d):.4f}, max {d.max():.3f}, logit sigma {l0.std():.2f}")
    assert np.median(d) < 0.03 and np.percentile(d, 99.9) < 0.4 and d.max() < 1.5   # the max sits on a crop with a content bit inside its soft knee
    assert (i0 == i1).mean() > 0.97


def test_token_prologue_in_self_kv_gemm_matches_separate_kernels(eng_bf16):
    """gemm_sk's token prologue (argmax of the previous step's logits + text_embed + pos_queries + norm_c inside the self_kv GEMM)
    vs argmax_kernel + dec_embed_ln_kernel + GEMM: the argmax is exact (first maximal index), the embedding sum is exact, only the
    fp32 reduction order of the LayerNorm differs.  45 rows: ragged 32-row tile."""
    rng = np.random.default_rng(14)
    crops = rng.integers(0, 256, (45, 32, 128, 3), dtype=np.uint8)
    try:
        assert eng_bf16.set_tuning(b"tok_fuse", 0) == 0
        l0, a0, i0 = eng_bf16.parseq_logits(crops, want_ar=True)
        assert eng_bf16.set_tuning(b"tok_fuse", 1) == 0
        l1, a1, i1 = eng_bf16.parseq_logits(crops, want_ar=True)
        l2, i2 = eng_bf16.parseq_logits(crops, want_ar=False)       # 25 AR steps: the 26th token still comes from the prologue
    finally:
        eng_bf16.set_tuning(b"tok_fuse", 1)
    assert np.isfinite(a1).all()
    assert np.abs(a1[:, 0] - a0[:, 0]).max() < 0.05                    # step 0: BOS for every crop; bf16 boundary flips only
    same_path = _same_path(a0, a1)
    assert same_path.mean() >= 0.9
    assert np.percentile(_ar_diff(a0, a1, same_path), 99.9) < 0.25 and _ar_diff(a0, a1, same_path).max() < 1.5
    assert np.percentile(np.abs(l1[same_path] - l0[same_path]), 99.9) < 0.25 and np.abs(l1[same_path] - l0[same_path]).max() < 1.5
    from tests.parity_rules import upto_eos
    keep = np.arange(26)[None, :] < upto_eos(i0)[:, None]
    assert np.array_equal(i0[same_path][keep[same_path]], i1[same_path][keep[same_path]])
    assert np.array_equal(l1, l2) and np.array_equal(i1, i2)            # the AR-logit output buffer does not change the result


@pytest.mark.parametrize("n", [45, 3])
def test_fused_mlp_block_matches_separate_kernels(eng_bf16, n):
    """mlp_fused.hip (norm2 + fc1 + GELU + fc2 + residual + next LayerNorm in one kernel, hidden activation in registers) vs
    layernorm_kernel + gemm_ws + gemm2: same rounding points, fp32 summation order differs.  45 crops = 45 row panels;
    3 crops = fewer panels than CUs."""
    rng = np.random.default_rng(13)
    crops = rng.integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
    try:
        assert eng_bf16.set_tuning(b"mlp_fused", 0) == 0
        l0, a0, i0 = eng_bf16.parseq_logits(crops, want_ar=True)
        assert eng_bf16.set_tuning(b"mlp_fused", 2) == 0          # 2: also below the row count where it pays
        l1, a1, i1 = eng_bf16.parseq_logits(crops, want_ar=True)
        l2, a2, i2 = eng_bf16.parseq_logits(crops, want_ar=True)
    finally:
        eng_bf16.set_tuning(b"mlp_fused", 1)
    assert np.isfinite(a1).all() and np.isfinite(l1).all()
    assert np.array_equal(a1, a2) and np.array_equal(l1, l2)              # run to run identical (no race in the weight ring)
    # A changed fp32 summation order anywhere in the 12-block encoder moves these random-noise crops' logits by ~0.25 (the same
    # spread as between the two GEMM kernel generations); the kernel's own accuracy is pinned in test_gpu_mlp.py.
    d0 = np.abs(a1[:, 0] - a0[:, 0])                                      # step 0: no token feedback yet
    assert np.median(d0) < 0.1 and np.percentile(d0, 99) < 0.8            # the tail: crops with a content bit inside its soft knee
    same_path = _same_path(a0, a1)
    assert same_path.mean() >= 0.9
    assert np.percentile(_ar_diff(a0, a1, same_path), 99.9) < 0.8


@pytest.mark.parametrize("n", [200, 400])
def test_fused_encoder_block_kernels_at_their_batch_sizes(eng_bf16, n):
    """Default kernel selection at 200 crops (qkv_attn on, mlp_fused off) and 400 crops (both on) against the layer-per-kernel
    encoder (both knobs off): same pipeline up to fp32 summation order (see the note on random-noise crops above)."""
    rng = np.random.default_rng(n)
    crops = rng.integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
    try:
        eng_bf16.set_tuning(b"mlp_fused", 0); eng_bf16.set_tuning(b"qkv_attn", 0)
        l0, a0, i0 = eng_bf16.parseq_logits(crops, want_ar=True)
        eng_bf16.set_tuning(b"mlp_fused", 1); eng_bf16.set_tuning(b"qkv_attn", 1)
        l1, a1, i1 = eng_bf16.parseq_logits(crops, want_ar=True)
        l2, a2, i2 = eng_bf16.parseq_logits(crops, want_ar=True)
    finally:
        eng_bf16.set_tuning(b"mlp_fused", 1); eng_bf16.set_tuning(b"qkv_attn", 1)
    assert np.isfinite(a1).all() and np.isfinite(l1).all()
    assert np.array_equal(a1, a2) and np.array_equal(l1, l2)
    d0 = np.abs(a1[:, 0] - a0[:, 0]).max(1)
    assert np.median(d0) < 0.2 and np.percentile(d0, 95) < 0.8
    same_path = _same_path(a0, a1)
    assert same_path.mean() >= 0.9
    assert np.percentile(_ar_diff(a0, a1, same_path), 99.9) < 0.8


@pytest.mark.parametrize("n", [45, 700])
def test_ar_early_exit_is_invisible_in_the_refined_logits(eng_bf16, n):
    """Upstream PARSeq leaves its AR loop once every crop of the batch has emitted EOS (system.py; SURVEY 2.2: the refined logits do
    not depend on it).  The engine does the same on the device - the per-step kernels return at once when the batch's done counter
    has reached N - so the steps behind the exit cost a launch boundary each instead of a decode step.  Keys behind a crop's EOS are
    masked in the refinement pass, so the refined logits must be BIT-IDENTICAL with and without the exit; the AR logits agree on
    every step that ran - per crop up to its EOS - and read zero behind the exit."""
    crops = np.random.default_rng(21).integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
    try:
        assert eng_bf16.set_tuning(b"ar_early_exit", 0) == 0
        l0, a0, i0 = eng_bf16.parseq_logits(crops, want_ar=True)
        assert eng_bf16.set_tuning(b"ar_early_exit", 1) == 0
        l1, a1, i1 = eng_bf16.parseq_logits(crops, want_ar=True)
        l2, i2 = eng_bf16.parseq_logits(crops)
    finally:
        eng_bf16.set_tuning(b"ar_early_exit", 1)
    assert np.array_equal(l0, l1) and np.array_equal(i0, i1) and np.array_equal(l1, l2)
    ran = np.abs(a1).max((0, 2)) > 0                       # steps that ran for the batch
    steps = int(ran.sum())
    assert ran[:steps].all() and not ran[steps:].any()
    from tests.parity_rules import upto_eos
    # per crop too: behind a crop's own EOS the step's two attention kernels skip it (ar_crop_exit), so its AR logits agree up to
    # and including the EOS step and are undefined (never read) behind it
    live = np.arange(26)[None, :] < upto_eos(a0.argmax(-1))[:, None]
    assert np.array_equal(a0[live], a1[live])
    longest = int(upto_eos(a0.argmax(-1)).max())
    print(f"AR early exit, {n} crops: {steps} of 26 steps ran; the longest string of the batch ends at step {longest}")
    assert steps <= longest + 1 < 26                       # the designed weights decode strings of at most 10 characters


@pytest.mark.parametrize("n", [37, 700])
def test_ar_tail_in_the_fused_kernel_matches_the_per_op_steps(eng_bf16, n):
    """With the early exit, AR steps from `ar_tail_step` on run as one launch of dec_fused.hip in its tail form (it picks up the
    tokens and the K/V cache of the kernel-per-op steps, and returns at once when the batch is done).  Forced to start at step 3,
    where most crops are still decoding, it must continue the same greedy paths as the kernel-per-op loop (fp32 summation order
    differs: engine-vs-engine bounds as for the full fused kernel)."""
    crops = np.random.default_rng(23).integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
    try:
        assert eng_bf16.set_tuning(b"ar_tail_step", 0) == 0         # kernel-per-op all the way
        l0, a0, i0 = eng_bf16.parseq_logits(crops, want_ar=True)
        assert eng_bf16.set_tuning(b"ar_tail_step", 3) == 0
        l1, a1, i1 = eng_bf16.parseq_logits(crops, want_ar=True)
        l2, i2 = eng_bf16.parseq_logits(crops)
    finally:
        eng_bf16.set_tuning(b"ar_tail_step", 12)
    assert np.isfinite(l1).all() and np.isfinite(a1).all()
    assert np.array_equal(a0[:, :3], a1[:, :3])                     # the steps before the hand-over are the same launches
    same_path = _same_path(a0, a1)
    assert same_path.mean() >= 0.9
    from tests.parity_rules import upto_eos
    keep = np.arange(26)[None, :] < upto_eos(a0.argmax(-1))[:, None]
    d = np.abs(a1 - a0)[same_path[:, None] & keep]
    assert np.percentile(d, 99.9) < 0.25 and d.max() < 1.5
    keep_r = np.arange(26)[None, :] < upto_eos(i0)[:, None]
    assert np.array_equal(i0[same_path][keep_r[same_path]], i1[same_path][keep_r[same_path]])
    assert np.array_equal(l1, l2) and np.array_equal(i1, i2)
