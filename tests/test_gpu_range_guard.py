"""-m gpu: the range guard of the default precision (tuatara_amd/csrc/split.h: RangeWatch).

f16x4 writes every activation as f16 planes: exact for |x| < 65504, silently saturated beyond.  Synthetic weights never get near the limit; the archives
the reference loads (/root/reference/tuatara.cpp:333, :423) are unknown.  Every kernel that writes planes therefore watches the values it splits, and a
batch whose activations left the range FAILS, naming the layer - instead of returning finite, wrong boxes and strings.  Here one layer's weights are
scaled until it trips, on CRAFT and on PARSeq; the other modes of the knob (warn, off) and the untripped engine are checked beside it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(d):
    from tests.conftest import _engine as E
    return E(d, "f16x4")


@pytest.fixture(scope="module")
def scaled_dirs(tmp_path_factory):
    """Two weight directories: CRAFT's slice2.14 scaled by 3e5 (its ReLU outputs pass 65504), PARSeq's encoder.blocks.3 fc1 scaled by 4e4."""
    from tuatara_amd import weights as W
    c, p = W.synth_craft(0, False), W.synth_parseq(0)
    dc = str(tmp_path_factory.mktemp("w_craft_hot"))
    c2 = dict(c)
    name = [conv for nm, conv, bn, cin, cout, k in W.craft_layers() if nm == "slice2.14"][0]
    c2[name + ".weight"] = c[name + ".weight"] * np.float32(3e5)
    W.export_craft(c2, dc)
    W.export_parseq(p, dc)
    dp = str(tmp_path_factory.mktemp("w_parseq_hot"))
    p2 = dict(p)
    for k in ("weight", "bias"):
        p2[f"encoder.blocks.3.mlp.fc1.{k}"] = p[f"encoder.blocks.3.mlp.fc1.{k}"] * np.float32(4e4)
    W.export_craft(c, dp)
    W.export_parseq(p2, dp)
    dk = str(tmp_path_factory.mktemp("w_parseq_hot_keys"))   # the decoder's cross-attention keys (rows 384 .. 767 of its in_proj) scaled by 1e5
    p3 = dict(p)
    for k in ("in_proj_weight", "in_proj_bias"):
        a = p[f"decoder.layers.0.cross_attn.{k}"].copy()
        a[384:768] *= np.float32(1e5)
        p3[f"decoder.layers.0.cross_attn.{k}"] = a
    W.export_craft(c, dk)
    W.export_parseq(p3, dk)
    return {"craft": dc, "parseq": dp, "parseq_keys": dk}


def test_craft_layer_out_of_range_fails_the_call_and_names_the_layer(scaled_dirs):
    from tuatara_amd.engine import EngineError
    eng = _engine(scaled_dirs["craft"])
    canvas = np.random.default_rng(0).integers(0, 256, (256, 192, 3), dtype=np.uint8)
    with pytest.raises(EngineError) as ei:
        eng.craft_heatmap(canvas)
    msg = str(ei.value)
    print(msg)
    assert "range guard" in msg and "craft.slice2.14" in msg and "65504" in msg
    # the word is cleared when it is reported: the next call reports again (same weights), it does not inherit
    with pytest.raises(EngineError):
        eng.craft_heatmap(canvas)
    # through the whole path: image_to_data fails too (the reference's convention one level up: the shim prints and returns [])
    page = np.random.default_rng(1).integers(0, 256, (200, 180, 3), dtype=np.uint8)
    with pytest.raises(EngineError) as ei:
        eng.image_to_data(page)
    assert "craft.slice2.14" in str(ei.value)
    # warn only: the (saturated) result comes back; off: silent
    assert eng.set_tuning(b"range_guard", 2) == 0
    h = eng.craft_heatmap(canvas)
    assert h.shape == (128, 96, 2)
    assert eng.set_tuning(b"range_guard", 0) == 0
    eng.craft_heatmap(canvas)
    assert eng.set_tuning(b"range_guard", 1) == 0
    with pytest.raises(EngineError):
        eng.craft_heatmap(canvas)


def test_parseq_layer_out_of_range_fails_the_call_and_names_the_layer(scaled_dirs):
    from tuatara_amd.engine import EngineError
    eng = _engine(scaled_dirs["parseq"])
    crops = np.random.default_rng(2).integers(0, 256, (40, 32, 128, 3), dtype=np.uint8)
    with pytest.raises(EngineError) as ei:
        eng.parseq_logits(crops)
    msg = str(ei.value)
    print(msg)
    assert "range guard" in msg and "encoder.blocks.3." in msg and "fc1" in msg
    # the detector of this directory is healthy: it does not trip
    canvas = np.random.default_rng(0).integers(0, 256, (256, 192, 3), dtype=np.uint8)
    assert np.isfinite(eng.craft_heatmap(canvas)).all()
    # a large batch goes through the 128-row tiles (other kernels than a page's 64-row ones): same verdict
    crops = np.random.default_rng(3).integers(0, 256, (300, 32, 128, 3), dtype=np.uint8)
    with pytest.raises(EngineError) as ei:
        eng.parseq_logits(crops)
    assert "encoder.blocks.3." in str(ei.value)


def test_cross_attention_keys_out_of_range_trip_where_they_become_f16(scaled_dirs):
    """The memory's K / V leave the cross_kv linear as fp32 rows; the AR steps' per-row cross-attention multiplies them in fp32, the refinement pass's
    matrix-core kernel (attn_cross_split.hip) splits them into f16 planes - and watches them there.  Keys of ~1e6: the call fails naming the decoder's
    cross-attention; with that kernel switched off (the vector form keeps K in fp32) the same weights go through."""
    from tuatara_amd.engine import EngineError
    eng = _engine(scaled_dirs["parseq_keys"])
    crops = np.random.default_rng(4).integers(0, 256, (24, 32, 128, 3), dtype=np.uint8)
    with pytest.raises(EngineError) as ei:
        eng.parseq_logits(crops)
    msg = str(ei.value)
    print(msg)
    assert "range guard" in msg and "decoder.cross_attn" in msg
    assert eng.set_tuning(b"cross_split", 0) == 0
    try:
        got, ids = eng.parseq_logits(crops)
    finally:
        eng.set_tuning(b"cross_split", 1)
    assert np.isfinite(got).all()


def test_healthy_weights_do_not_trip_and_cost_nothing_visible(eng_x4):
    """The shipped engine on the benchmark's weights: no trip on a page and on a crop batch; results identical with the guard on and off (it only watches)."""
    crops = np.random.default_rng(5).integers(0, 256, (64, 32, 128, 3), dtype=np.uint8)
    a, ida = eng_x4.parseq_logits(crops)
    assert eng_x4.set_tuning(b"range_guard", 0) == 0
    try:
        b, idb = eng_x4.parseq_logits(crops)
    finally:
        eng_x4.set_tuning(b"range_guard", 1)
    assert np.array_equal(a, b) and np.array_equal(ida, idb)


def test_non_finite_weights_are_refused_at_load(tmp_path):
    """The guard sees magnitudes (a maximum): a NaN slips under it.  A NaN can only come from an infinity (caught) or from the file - refused by name."""
    from tuatara_amd import weights as W
    from tuatara_amd.engine import Engine, EngineError
    c, p = W.synth_craft(0, True), W.synth_parseq(0)
    p2 = dict(p)
    w = p["decoder.layers.0.linear1.weight"].copy()
    w[7, 9] = np.nan
    p2["decoder.layers.0.linear1.weight"] = w
    W.export_craft(c, str(tmp_path))
    W.export_parseq(p2, str(tmp_path))
    with pytest.raises(EngineError) as ei:
        Engine(str(tmp_path))
    assert "non-finite" in str(ei.value) and "linear1" in str(ei.value)


@pytest.fixture(scope="module")
def first_layer_hot_dir(tmp_path_factory):
    """CRAFT's first convolution scaled by 1e6 (weights only): a black page passes (its output is the folded bias), any page with light on it leaves the f16
    range in 'craft.slice1.0' - a detector trip that depends on the INPUT, so that one batch of a stream can fail between two that do not."""
    from tuatara_amd import weights as W
    c, p = W.synth_craft(0, False), W.synth_parseq(0)
    d = str(tmp_path_factory.mktemp("w_craft_first_hot"))
    c2 = dict(c)
    name = [conv for nm, conv, bn, cin, cout, k in W.craft_layers() if nm == "slice1.0"][0]
    c2[name + ".weight"] = c[name + ".weight"] * np.float32(1e6)
    W.export_craft(c2, d)
    W.export_parseq(p, d)
    return d


def test_streamed_batches_the_offending_batch_fails_and_its_neighbours_survive(first_layer_hot_dir):
    """with streamed batches the detector of batch j and the recogniser of batch j - 1 are on the GPU together; one shared sticky word blamed
    the wrong batch.  Now every stage and slot has its own word (engine.h: kRangeDet0 ...): the detector's is verified in detect_collect, before the batch's boxes
    are used.  Three batches through ttr_stream_push - black pages, pages of noise, black pages: the push of the noise batch fails naming craft.slice1.0, the
    batches before and after it come back, and the engine is usable afterwards."""
    from tuatara_amd.engine import DeviceBuffer, EngineError
    eng = _engine(first_layer_hot_dir)
    H, W_, n = 256, 192, 2
    black = np.zeros((n, H, W_, 3), np.uint8)
    noise = np.random.default_rng(7).integers(0, 256, (n, H, W_, 3), dtype=np.uint8)
    bufs = []
    for a in (black, noise, black):
        d = DeviceBuffer(a.nbytes)
        d.upload(a)
        bufs.append(d)
    assert eng.stream_push(bufs[0], n, H, W_) == []                    # batch 0 in
    with pytest.raises(EngineError) as ei:
        eng.stream_push(bufs[1], n, H, W_)                             # batch 1: its own detector trips, it never enters the pipeline
    msg = str(ei.value)
    print(msg)
    assert "range guard" in msg and "craft.slice1.0" in msg and "detector" in msg
    got = []
    r = eng.stream_push(bufs[2], n, H, W_)                             # batch 2: healthy
    if r:
        got.append(r)
    for _ in range(3):
        r = eng.stream_flush()
        if not r:
            break
        got.append(r)
    assert len(got) == 2 and all(len(b) == n for b in got), [len(b) for b in got]      # batches 0 and 2, n pages each
    sync = eng.pages_to_data_dev(bufs[0], n, H, W_)                    # nothing left in flight: the synchronous call runs - and gives what the streamed batches gave
    same = lambda a, b: len(a) == len(b) and all(x["text"] == y["text"] and x["bbox"] == y["bbox"] for x, y in zip(a, b))
    assert all(same(p, q) for b in got for p, q in zip(b, sync))
    with pytest.raises(EngineError):
        eng.pages_to_data_dev(bufs[1], n, H, W_)
    # the list form (ttr_images_to_data): the noise image's batch fails, the black images are delivered, the call reports which
    res = eng.images_to_data([black[0], np.ascontiguousarray(noise[0].transpose(1, 0, 2)), black[1], np.zeros((64, 64, 3), np.uint8)])   # (the noise image 192 x 256: a size, hence a batch, of its own)
    assert len(res) == 4 and res[1] == [] and same(res[0], sync[0]) and same(res[2], sync[1])
    assert eng.last_images_error and "craft.slice1.0" in eng.last_images_error and " 1" in eng.last_images_error.split("):")[0], eng.last_images_error
    assert "1 of 4 images failed" in eng.last_images_error
    for d in bufs:
        d.free()


def test_stage_entry_points_refuse_while_batches_are_in_flight(eng_x4):
    """ttr_parseq_logits / ttr_craft_heatmap share workspaces with the batches; with the recogniser of a streamed batch on a stream of its
    own they would race with it.  They refuse, like the synchronous page calls, until ttr_stream_flush has returned everything."""
    from tuatara_amd.engine import DeviceBuffer, EngineError
    pages = np.full((1, 128, 160, 3), 255, np.uint8)
    d = DeviceBuffer(pages.nbytes)
    d.upload(pages)
    eng_x4.stream_push(d, 1, 128, 160)
    crops = np.zeros((2, 32, 128, 3), np.uint8)
    try:
        with pytest.raises(EngineError, match="in flight"):
            eng_x4.parseq_logits(crops)
        with pytest.raises(EngineError, match="in flight"):
            eng_x4.craft_heatmap(np.zeros((64, 64, 3), np.uint8))
    finally:
        while eng_x4.stream_flush():
            pass
    eng_x4.parseq_logits(crops)
    d.free()
