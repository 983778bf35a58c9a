"""-m gpu: the bf16 (benchmarked) mode against the CPU fp32 oracle at the sizes the benchmark runs — BASELINE.json configs 2
and 5 — under the margin rule of tests/parity_rules.py.  At these sizes the engine picks its large-batch kernels (qkv_attn
from 160 crops, mlp_fused from 384, the fused refinement block and MFMA cross-attention), which the 24/48-crop tests never
reach."""
import numpy as np
import pytest

from tests import parity_rules as R

pytestmark = pytest.mark.gpu


def _oracle_logits(parseq, crops, batch=64):
    return R.oracle_logits(parseq, crops, batch)      # (memoised: tests/parity_rules.py)


@pytest.mark.parametrize("n", [256, 448])
def test_bf16_parseq_vs_oracle_at_bench_batch_sizes(eng_bf16, oracle_models, n):
    """Config 2 (256 random crops, seed 0: qkv_attn active) and 448 crops (qkv_attn + mlp_fused active): bf16 engine vs the CPU
    oracle.  >= 90 % of the crops decode along the oracle's greedy path, no divergence at a confident position."""
    _, parseq = oracle_models
    crops = np.random.default_rng(0).integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
    ref, ref_ar = _oracle_logits(parseq, crops)
    got, got_ar, ids = eng_bf16.parseq_logits(crops, want_ar=True)
    assert np.isfinite(got).all() and np.isfinite(got_ar).all()
    st = R.parseq_margin_rule(ref, ref_ar, got, got_ar, min_same=0.9, label=f"bf16 PARSeq, {n} crops vs oracle")
    assert st["mean_dlogit"] < 0.5
    from oracle import post
    from tuatara_amd.engine import decode_ids
    s_ref, _ = post.decode_logits(ref)
    s_got = [decode_ids(r) for r in ids]
    same = st["same_mask"]
    assert all(a == b for a, b, s in zip(s_got, s_ref, same) if s)          # identical strings wherever the path is the oracle's
    print(f"   strings identical on {sum(a == b for a, b in zip(s_got, s_ref))}/{n} crops; {len(set(s_ref))} distinct strings")


def test_f32_parseq_256_crops_within_1e3(eng_f32, oracle_models):
    """Config 2 in parity mode at its full size: logits within north_star's 1e-3, identical ids."""
    _, parseq = oracle_models
    crops = np.random.default_rng(0).integers(0, 256, (256, 32, 128, 3), dtype=np.uint8)
    ref, ref_ar = _oracle_logits(parseq, crops)
    got, got_ar, ids = eng_f32.parseq_logits(crops, want_ar=True)
    up = R.upto_eos(ref.argmax(-1))
    mask = np.arange(26)[None, :] < up[:, None]
    live = np.arange(26)[None, :] < R.upto_eos(ref_ar.argmax(-1))[:, None]      # AR logits are defined up to each crop's EOS (early exit)
    assert np.array_equal(got_ar.argmax(-1)[live], ref_ar.argmax(-1)[live])
    err = np.abs(got - ref)
    print(f"f32 PARSeq 256 crops: max |dlogit| {err.max():.2e} (up to EOS {err[mask].max():.2e}); logit max {np.abs(ref).max():.1f}")
    assert err[mask].max() < 1e-3 and np.abs(got_ar - ref_ar)[mask].max() < 1e-3
    assert np.array_equal(ids[mask], ref.argmax(-1)[mask])


def _pages(n, words=28):
    from tuatara_amd import synth
    return [synth.synthetic_page(i, 1024, 768, n_words=words) for i in range(n)]


def _stage_path(eng, page):
    canvas, ratio = eng.resize_canvas(page)                                       # tuatara.cpp:349 swap + :206-234 inside
    heat = eng.craft_heatmap(canvas)
    rects = eng.ccl_boxes(heat)
    crops, boxes = eng.pack_crops(page, rects, ratio)
    return heat, rects, crops, boxes


def test_bf16_config5_32_pages_vs_oracle_and_f32(eng_bf16, eng_f32, oracle_models):
    """Config 5 at the benchmark's step size: 32 synthetic 1024x768 pages (~1200 crops) in one batch.
      (1) f32 engine == CPU oracle on pages 0..1 (boxes and strings identical) - the f32 engine is then the reference for all 32;
      (2) bf16 batch == bf16 page by page (boxes identical; strings identical up to margin-rule ties);
      (3) bf16 heat maps within a bounded error of f32's, boxes = reference post-processing of bf16's own heat map, >= 90 % of
          the f32 boxes reproduced exactly;
      (4) the ~1200 crops both modes agree on, through PARSeq in one batch in both modes: margin rule, f32 as reference; the
          first 96 of them also against the CPU oracle."""
    from oracle import pipeline, post
    from tuatara_amd.engine import DeviceBuffer, decode_ids
    craft, parseq = oracle_models
    P = 32
    pages = _pages(P)
    buf = DeviceBuffer(P * 1024 * 768 * 3)
    buf.upload(np.stack(pages))
    res_b = eng_bf16.pages_to_data_dev(buf, P, 1024, 768)
    res_f = eng_f32.pages_to_data_dev(buf, P, 1024, 768)
    n_crops = sum(len(r) for r in res_b)
    print(f"config 5: {P} pages, {n_crops} bf16 crops ({n_crops / P:.1f} per page), {sum(len(r) for r in res_f)} f32 crops")
    assert n_crops >= 30 * P
    # (1)
    for k in range(2):
        ref = pipeline.image_to_data(craft, parseq, pages[k])
        assert [g["bbox"] for g in res_f[k]] == [r["bbox"] for r in ref], k
        assert [g["text"] for g in res_f[k]] == [r["text"] for r in ref], k
    # (2)
    n_same_txt = n_txt = 0
    for k in range(4):
        one = eng_bf16.image_to_data(pages[k])
        assert [g["bbox"] for g in one] == [g["bbox"] for g in res_b[k]], k
        n_txt += len(one)
        n_same_txt += sum(a["text"] == b["text"] for a, b in zip(one, res_b[k]))
    print(f"config 5: bf16 batch vs page-by-page: boxes identical, {n_same_txt}/{n_txt} strings identical")
    assert n_same_txt >= 0.95 * n_txt
    # (3)
    exact = total = 0
    crops_all, flips_total, set_total = [], 0, 0
    for k in range(P):
        hb, rb, cb, bb = _stage_path(eng_bf16, pages[k])
        if k < 8:
            hf, rf, cf, bf_ = _stage_path(eng_f32, pages[k])
            flips, err, nset = R.heatmap_flips(hf, hb)
            flips_total += int(flips.sum())
            set_total += nset
            assert np.abs(hb - hf).max() < 0.08 and err < 0.08, (k, np.abs(hb - hf).max(), err)
            det, _, _ = post.get_detected_boxes(hb[..., 0], hb[..., 1])                  # reference post-processing, bf16 heat map
            assert np.array_equal(rb, det), k
        # the batch call's boxes are the stage path's boxes
        bbox_stage = [post.tesseract_bbox(b) for b in bb]
        assert bbox_stage == [g["bbox"] for g in res_b[k]], k
        fset = {tuple(g["bbox"]) for g in res_f[k]}
        keep = [i for i, b in enumerate(bbox_stage) if tuple(b) in fset]
        exact += len(keep)
        total += len(res_f[k])
        crops_all.append(cb[keep])
    print(f"config 5: {exact}/{total} f32 boxes reproduced exactly by bf16; heat-map threshold flips on pages 0..7: {flips_total} of {set_total} set pixels")
    assert exact >= 0.9 * total
    # (4)
    crops = np.concatenate(crops_all)
    lf, af, idf = eng_f32.parseq_logits(crops, want_ar=True)
    lb, ab, idb = eng_bf16.parseq_logits(crops, want_ar=True)
    st = R.parseq_margin_rule(lf, af, lb, ab, min_same=0.9, label=f"bf16 vs f32 engine, {len(crops)} config-5 crops in one batch")
    ro, ao = _oracle_logits(parseq, crops[:96])
    up = R.upto_eos(ro.argmax(-1))
    mask = np.arange(26)[None, :] < up[:, None]
    assert np.abs(lf[:96] - ro)[mask].max() < 1e-3                                      # f32 engine at this batch size == oracle
    R.parseq_margin_rule(ro, ao, lb[:96], ab[:96], min_same=0.85, label="bf16 engine vs CPU oracle, first 96 config-5 crops")
    s_b = [decode_ids(r) for r in idb]
    s_f = [decode_ids(r) for r in idf]
    print(f"config 5: strings identical bf16 vs f32 on {sum(a == b for a, b in zip(s_b, s_f))}/{len(crops)} crops")
    buf.free()
