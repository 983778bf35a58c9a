#!/usr/bin/env python3
"""Benchmark of the OCR hot path (BASELINE.json metric: pages/sec whole-node, 1024x768 pages,
~40 crops/page, + p50 page latency).

  python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (resize/pad -> CRAFT -> union-find CCL -> calipers ->
crop-batch packer -> PARSeq -> token ids) over one batch of `--pages` synthetic pages per GPU,
inputs already resident in HBM.  Page-level data parallelism: each rank owns its pages and a
full weights replica (weak scaling); for N > 1 the decoded token ids of every rank are
all-gathered with RCCL (torch.distributed backend "nccl") inside the timed region — the only
exchange the path has.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CRAFT_GFLOP_PER_PAGE = 559.5      # 27 convs, 2*MACs, BN folded, 1024x768
PARSEQ_GFLOP_PER_CROP = 6.129     # encoder 5.747 + KV-cached AR 0.190 + refine 0.191
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
MFMA_F32_PEAK_TFLOPS = 157.3


def cpu_baseline(pages, craft_state, parseq_state, wdir, n_pages: int = 4):
    """The CPU path the reference runs (LibTorch fp32 + OpenCV), restated by the oracle (torch fp32 + the C restatement of the
    OpenCV steps), timed on the box's host cores on a bounded sample of the same workload, under the two schedules of SURVEY.md
    section 8(d).  Reported beside the GPU number; it is not the target.

      reference_faithful  tuatara.cpp as written: both TorchScript archives loaded inside every call (:336, :428), the recogniser
                          in chunks of 4 crops (:452) on 6 threads sharing one module (:461-475)
      best_effort         models loaded once, all crops of a page in one batch

    Both run through the Python port (torch CPU ops are LibTorch's; the TorchScript archives of schedule (i) are traced from the
    same seeded weights and loaded with torch.jit.load, the reference's load path).  A C++ harness would need the OpenCV half of
    tuatara.cpp, which cannot be built here."""
    import queue
    import threading

    import numpy as np
    import torch

    from oracle import pipeline, post

    craft, parseq = pipeline.load_models(craft_state, parseq_state)
    sample = pages[:n_pages]
    H, Wd = sample[0].shape[:2]
    cpath, ppath = os.path.join(wdir, "craft_traced_torchscript_model.pt"), os.path.join(wdir, "parseq_torchscript.bin")
    with torch.no_grad():
        canvas, _ = post.resize_aspect_ratio(np.ascontiguousarray(sample[0][:, :, ::-1]))
        torch.jit.trace(craft, torch.zeros(1, 3, canvas.shape[0], canvas.shape[1]), check_trace=False).save(cpath)
        torch.jit.trace(parseq, torch.zeros(4, 3, 32, 128), check_trace=False).save(ppath)

    def faithful(img):
        det = torch.jit.load(cpath)                                           # :333-336, per call
        d = pipeline.detect(det, img)
        crops = [c for c in (post.crop_resize(d["swapped"], b, True) for b in d["boxes"]) if c is not None]
        rec = torch.jit.load(ppath)                                           # :423-428, per call
        q, outs, lock = queue.Queue(), [], threading.Lock()
        for i in range(0, len(crops), 4):                                     # :450-459
            ch = np.stack(crops[i:i + 4])
            n = len(ch)
            if n < 4:                                                         # the traced archive has a fixed batch of 4: pad the last chunk
                ch = np.concatenate([ch, np.repeat(ch[-1:], 4 - n, 0)])
            q.put((i, n, torch.from_numpy(ch).permute(0, 3, 1, 2).float().div(255.0)))

        def infer():                                                          # :289-312
            while True:
                try:
                    i, n, x = q.get_nowait()
                except queue.Empty:
                    return
                with torch.no_grad():
                    y = rec(x)[:n]
                with lock:
                    outs.append((i, y))

        th = [threading.Thread(target=infer) for _ in range(6)]               # :461-475
        for t in th:
            t.start()
        for t in th:
            t.join()
        outs.sort(key=lambda t: t[0])                                         # :478
        logits = torch.cat([y for _, y in outs]).softmax(-1).numpy() if outs else np.zeros((0, 26, 95), np.float32)   # :485-486
        return post.decode_logits(logits)[0]

    pipeline.image_to_data(craft, parseq, sample[0][:256, :256].copy())       # warm-up (allocator, threads)
    t0 = time.perf_counter()
    ncrops = sum(len(pipeline.image_to_data(craft, parseq, pg)) for pg in sample)
    dt_best = time.perf_counter() - t0
    t0 = time.perf_counter()
    n_f = sum(len(faithful(pg)) for pg in sample[:max(2, n_pages // 2)])
    dt_faith = time.perf_counter() - t0
    nf_pages = max(2, n_pages // 2)
    return {"value": len(sample) / dt_best, "unit": "pages/s", "cores": torch.get_num_threads(), "kind": "port", "nproc": os.cpu_count(),
            "sample": f"{len(sample)} of the benchmark's synthetic {H}x{Wd} pages ({ncrops} crops as detected by the oracle), best-effort schedule: models loaded once, "
                      f"one PARSeq batch per page, torch {torch.__version__} fp32, {dt_best:.1f} s",
            "schedules": {"best_effort": {"pages_per_s": len(sample) / dt_best, "pages": len(sample), "seconds": dt_best},
                          "reference_faithful": {"pages_per_s": nf_pages / dt_faith, "pages": nf_pages, "crops": n_f, "seconds": dt_faith,
                                                 "what": "TorchScript archives loaded per call, PARSeq in chunks of 4 on 6 threads (tuatara.cpp:336, :428, :452, :461)"}},
            "torch_threads": torch.get_num_threads(), "implementation": "Python port (oracle/): torch CPU fp32 + C restatement of the OpenCV steps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pages", type=int, default=32, help="pages per GPU per step (CRAFT runs in groups of 16, PARSeq over all crops of the step)")
    ap.add_argument("--words", type=int, default=40, help="words drawn per synthetic page (SURVEY.md section 8d: ~40 random words)")
    ap.add_argument("--boxes", default="grid40", choices=["grid40", "detected"], help="grid40 (SURVEY.md section 8d): CRAFT + CCL + box extraction run in full (timed), then every page's "
                    "boxes are replaced by a fixed 5 x 8 grid of 150 x 40 px boxes so that PARSeq sees exactly 40 crops per page; detected: the synthetic detector's own boxes")
    ap.add_argument("--buffers", type=int, default=16, help="distinct device page buffers rotated over the steps (each holds --pages distinct pages of the 512-seed stream; 16 x 32 = all 512 seeds)")
    ap.add_argument("--parity-pages", type=int, default=8, help="pages per step of the f32 parity-mode measurement after the timed region (0 = skip)")
    ap.add_argument("--contexts", type=int, default=1, help="engine contexts (HIP streams + host threads) per GPU; a step's pages are split between them")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32", "f16x4"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-iters", type=int, default=20)
    ap.add_argument("--stream", type=int, default=1, help="1: feed the steps through ttr_stream_push (batch j's detector and batch j-1's recogniser are enqueued before batch j-2's results are awaited, "
                    "host box extraction overlaps GPU work; one stream, kernels still run alone); 0: one synchronous ttr_pages_to_data_dev call per step")
    ap.add_argument("--tune", action="append", default=[], help="engine tuning knob key=value (ttr_set_tuning), repeatable")
    ap.add_argument("--decoder-mode", type=int, default=None, help="ttr_set_decoder_mode override (0 = kernel per op, 4/8/16 = fused)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N")
        args.gpus = world

    import numpy as np
    import torch

    from tuatara_amd import build as B
    from tuatara_amd import synth
    from tuatara_amd import weights as W
    from tuatara_amd.engine import DeviceBuffer, Engine

    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if rank == 0:
        B.build_lib()
    if dist:
        dist.barrier()
    wdir = os.path.join(tempfile.gettempdir(), f"tuatara_bench_weights_{os.getuid()}_{local_rank}")
    craft_state, parseq_state = W.make_synthetic_weights(wdir, seed=0, structured=True)
    from concurrent.futures import ThreadPoolExecutor

    P, H, Wd = args.pages, 1024, 768
    NC = max(1, min(args.contexts, P))
    grid = 1 if args.boxes == "grid40" else 0
    engs = [Engine(wdir, precision=args.precision, device=local_rank, bench_grid_boxes=grid) for _ in range(NC)]
    eng = engs[0]
    if args.decoder_mode is not None:
        eng.set_tuning(b"decoder_mode", args.decoder_mode)
    for kv in args.tune:
        k, v = kv.split("=")
        assert eng.set_tuning(k.encode(), int(v)) == 0, kv
    # the 512-seed stream of SURVEY.md section 8d: step k of rank r works on pages (r * NB + k % NB) * P .. + P - 1 (mod 512), NB distinct
    # device-resident buffers rotated so that consecutive steps never see the same pages
    NB = max(3, args.buffers)
    seeds = [[(((rank * NB + b) * P + i) % 512) for i in range(P)] for b in range(NB)]
    host_pages = [[synth.synthetic_page(sd, H, Wd, n_words=args.words) for sd in seeds[b]] for b in range(NB)]
    pages = host_pages[0]
    # each context owns a contiguous share of the step's pages, resident in HBM before the timed region
    share = [P // NC + (1 if c < P % NC else 0) for c in range(NC)]
    first = [sum(share[:c]) for c in range(NC)]
    dbufs = [[None] * NC for _ in range(NB)]
    for b in range(NB):
        for c in range(NC):
            d = DeviceBuffer(share[c] * H * Wd * 3)
            d.upload(np.stack(host_pages[b][first[c]:first[c] + share[c]]))
            dbufs[b][c] = d
    step_no = [0]
    pool = ThreadPoolExecutor(max_workers=NC)

    from tuatara_amd import dist as D

    def step():
        # the C ABI call releases the GIL: the contexts' host work (calipers, launches) and GPU work overlap
        bsel = step_no[0] % NB
        step_no[0] += 1
        futs = [pool.submit(engs[c].pages_to_data_dev, dbufs[bsel][c], share[c], H, Wd) for c in range(NC)]
        res = [r for f in futs for r in f.result()]
        if dist:  # fixed-size records (<=128 crops x 26 token ids per page) gathered over RCCL/xGMI
            D.all_gather_records(D.pack_records(res), device="cuda")
        return res

    stream = bool(args.stream) and NC == 1

    def run_steps(k_steps):
        """k_steps whole steps; returns the last step's results.  Streamed: every step's results come back two pushes later, the last
        two from the flushes — all inside the caller's timed region."""
        if not stream:
            out = None
            for _ in range(k_steps):
                out = step()
            return out
        out = None
        for _ in range(k_steps):
            bsel = step_no[0] % NB                # a buffer is pushed again NB >= 2 pushes later: its results came back one push before
            step_no[0] += 1
            prev = eng.stream_push(dbufs[bsel][0], P, H, Wd)
            if prev:
                if dist:
                    D.all_gather_records(D.pack_records(prev), device="cuda")
                out = prev
        while True:                      # the (up to two) batches still in flight
            last = eng.stream_flush()
            if not last:
                break
            if dist:
                D.all_gather_records(D.pack_records(last), device="cuda")
            out = last
        return out

    res = run_steps(args.warmup) if args.warmup else None
    crops_per_page = float(np.mean([len(r) for r in res])) if args.warmup else 0.0

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        for e in engs:
            e.lib.ttr_dev_sync(e.h)

    for e in engs:
        e.set_profiling(1)     # timed region: HIP events around the dominant kernels only (the CRAFT convolutions)
    fence()
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    prof = {k: {"ms": 0.0, "flops": 0.0, "launches": 0} for k in ("craft", "parseq", "parseq_ar")}
    for e in engs:
        pe = e.get_profile()
        for k in prof:
            for f in prof[k]:
                prof[k][f] += pe[k][f]
        e.set_profiling(False)
    stage = eng.last_stage_ms()
    crops_per_page = float(np.mean([len(r) for r in res]))
    # secondary rooflines (ViT / decoder GEMMs): two more steps with every launch bracketed by events, outside the timed
    # region (1400 event records per step cost ~8 % of throughput, so they stay out of `value`)
    for e in engs:
        e.set_profiling(2)
    run_steps(2)
    fence()
    for e in engs:
        pe = e.get_profile()
        for k in ("parseq", "parseq_ar"):
            for f in prof[k]:
                prof[k][f] += pe[k][f]
        e.set_profiling(0)
    SEC_STEPS = 2
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # p50 single-page latency (one page per call, synchronous) — outside the timed region
    lat = []
    one = DeviceBuffer(H * Wd * 3)
    one.upload(pages[0])
    for _ in range(args.latency_iters):
        t1 = time.perf_counter()
        eng.pages_to_data_dev(one, 1, H, Wd)
        lat.append((time.perf_counter() - t1) * 1e3)
    p50 = float(np.median(lat)) if lat else None

    # the same steps with the host -> device copy of every step's pages inside the timed span (pageable numpy -> HBM, synchronous
    # hipMemcpy before each push: the un-overlapped upper bound of what a caller holding host buffers pays) -- never `value`
    h2d_rate = None
    if rank == 0 or dist:
        k_h2d = max(2, min(4, args.steps))
        stacks = [np.stack(host_pages[b]) for b in range(NB)]
        fence()
        t1 = time.perf_counter()
        step_no[0] = 0
        for k in range(k_h2d):                          # upload, then push: buffer k % NB last went out NB >= 3 pushes ago, its results are back
            for c in range(NC):
                dbufs[k % NB][c].upload(stacks[k % NB][first[c]:first[c] + share[c]])
            if stream:
                prev = eng.stream_push(dbufs[k % NB][0], P, H, Wd)
                step_no[0] += 1
                if prev and dist:
                    D.all_gather_records(D.pack_records(prev), device="cuda")
            else:
                step()
        while stream:
            last = eng.stream_flush()
            if not last:
                break
            if dist:
                D.all_gather_records(D.pack_records(last), device="cuda")
        fence()
        h2d_rate = world * P * k_h2d / (time.perf_counter() - t1)

    # parity mode (f32: logits within 1e-3 of the CPU reference path, identical boxes and strings) on the same workload
    parity = None
    if rank == 0 and args.parity_pages > 0 and args.precision == "bf16":
        pe = Engine(wdir, precision="f32", device=local_rank, bench_grid_boxes=grid)
        pp = min(args.parity_pages, P)
        pb = DeviceBuffer(pp * H * Wd * 3)
        pb.upload(np.stack(host_pages[0][:pp]))
        pe.pages_to_data_dev(pb, pp, H, Wd)
        pe.lib.ttr_dev_sync(pe.h)
        t1 = time.perf_counter()
        for _ in range(2):
            rp = pe.pages_to_data_dev(pb, pp, H, Wd)
        pe.lib.ttr_dev_sync(pe.h)
        parity = {"pages_per_s": 2 * pp / (time.perf_counter() - t1), "pages_per_step": pp, "steps": 2, "crops_per_page": float(np.mean([len(r) for r in rp])),
                  "dtype": "f32", "what": "parity mode: every conv / linear on v_mfma_f32_16x16x4_f32 (exact fp32 products), the mode tests/ hold to 1e-3 against the oracle"}
        pe.close()
        pb.free()

    if rank == 0:
        # HBM bytes of the CRAFT conv kernels per launch, from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE,
        # gfx950 corrections applied; the json names the build it was taken on) -- counters cannot be read from inside this process
        traffic = traffic_src = None
        for name in ("r02_pmc_craft_b16_v2.json", "r02_pmc_craft_b16.json", "r01_pmc_craft_b16.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:   # measured on 16-page CRAFT groups
                    tj = json.load(f)
                traffic = tj["craft_conv_kernels"]["hbm_bytes_per_launch"] * min(P, 16) / 16.0
                traffic_src = {"file": "profiles/" + name, "build": tj.get("build")}
                break
            except Exception:
                pass
        total_pages = world * P * args.steps
        peak = MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16" else MFMA_F32_PEAK_TFLOPS
        c = prof["craft"]
        craft_tflops = (CRAFT_GFLOP_PER_PAGE * 1e9 * P * args.steps) / (c["ms"] * 1e-3) / 1e12 if c["ms"] else None
        q = prof["parseq"]
        pq_tflops = (q["flops"] / (q["ms"] * 1e-3) / 1e12) if q["ms"] else None
        out = {
            "metric": "pages/sec whole-node (1024x768, ~40 crops/page)", "value": total_pages / dt, "unit": "pages/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "configs[4]: synthetic stream of 1024x768 pages (~40 detected crops each), page-level DP, "
                                   "RCCL all-gather of token ids", "pages_per_gpu_per_step": P, "engine_contexts_per_gpu": NC, "batches_in_flight": 3 if stream else 1,
                       "words_drawn_per_page": args.words, "crops_per_page": round(crops_per_page, 1), "boxes": args.boxes,
                       "distinct_pages": NB * P, "page_buffers_rotated": NB, "weights": "seeded synthetic (designed read-outs on random CRAFT / PARSeq, tuatara_amd/weights.py)",
                       "parallelism": f"dp{world}"},
            "p50_page_latency_ms": p50,
            "h2d_included_pages_per_s": h2d_rate,
            "parity_mode": parity, "parity_mode_pages_per_s": parity["pages_per_s"] if parity else None,
            "stage_ms_last_step": {k: round(v, 3) for k, v in stage.items()},
            "roofline": {"kernel": "CRAFT convolutions: conv3p_first2s_kernel / conv3p_kernel / conv3s_kernel / gemm2_kernel (24 launches per 16-page group)", "bound": "mfma",
                         "achieved": craft_tflops, "peak": peak, "unit": "TFLOP/s",
                         "frac": (craft_tflops / peak) if craft_tflops else None, "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes over 16-page CRAFT groups)", "traffic_source": traffic_src,
                         "launches_per_step": c["launches"] / max(1, args.steps * NC), "avg_launch_us": c["ms"] * 1e3 / max(1, c["launches"]),
                         "algorithmic_gflop_per_page": CRAFT_GFLOP_PER_PAGE},
            "roofline_parseq_gemm": {"kernel": "PARSeq batched GEMM launches: gemm_ws_kernel (qkv, cross K/V), mlp_fused_kernel (proj + fc1 + fc2 of a block, LayerNorms and GELU included in its time), gemm2_kernel (patch embedding, refinement pass)", "bound": "mfma", "achieved": pq_tflops, "peak": peak,
                                     "unit": "TFLOP/s", "frac": (pq_tflops / peak) if pq_tflops else None,
                                     "launches_per_step": q["launches"] / SEC_STEPS, "measured": "2 extra steps after the timed region"},
            "roofline_parseq_ar_gemm": {"kernel": "gemm_sk_kernel: per-step autoregressive decoder linears (M = crops in flight; latency-bound)",
                                        "achieved": (prof["parseq_ar"]["flops"] / (prof["parseq_ar"]["ms"] * 1e-3) / 1e12) if prof["parseq_ar"]["ms"] else None,
                                        "unit": "TFLOP/s", "launches_per_step": prof["parseq_ar"]["launches"] / SEC_STEPS},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(pages, craft_state, parseq_state, wdir)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "pages/s", "cores": None, "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
