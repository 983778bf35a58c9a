"""GPU box: one synthetic 1024x768 page (grid40 boxes, 40 crops) through the synchronous single-page call: p50 latency, the host's wall-clock splits and the
GPU stage spans (HIP events), for a list of tuning settings - is a page's recogniser bound by the host's enqueue rate or by the GPU?
   python3 tools/page_host_balance.py "k=v k=v" "k=v" ..."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tuatara_amd import synth, weights as W
from tuatara_amd.engine import DeviceBuffer, Engine

d = tempfile.mkdtemp()
W.make_synthetic_weights(d, seed=0, structured=True)
eng = Engine(d, bench_grid_boxes=1)
page = synth.synthetic_page(0, 1024, 768, n_words=40, layout="cells5x8")
buf = DeviceBuffer(page.nbytes); buf.upload(page)
names = ["enqueue_detector", "wait_for_ccl", "copy_components", "calipers_and_boxes", "enqueue_recogniser", "wait_for_ids", "finish_misc", "decode_strings"]
for cfg in (sys.argv[1:] or [""]):
    sets = [kv.split("=") for kv in cfg.split()]
    for k, v in sets:
        assert eng.set_tuning(k.encode(), int(v)) == 0, k
    for _ in range(5):
        eng.pages_to_data_dev(buf, 1, 1024, 768)
    lat, hus, st = [], [], []
    for _ in range(30):
        t = time.perf_counter(); eng.pages_to_data_dev(buf, 1, 1024, 768); lat.append((time.perf_counter() - t) * 1e3)
        hus.append(eng.last_host_us()); st.append(eng.last_stage_ms())
    hu = np.median(np.array(hus), 0)
    sm = {k: float(np.median([s[k] for s in st])) for k in st[0]}
    print(f"[{cfg or 'default'}] p50 {np.median(lat):.3f} ms | host us: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, hu)) + f" | GPU stage ms: {sm}", flush=True)
    for k, v in sets:
        pass
