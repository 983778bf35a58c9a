#!/usr/bin/env python3
"""Benchmark of the OCR hot path (BASELINE.json metric: pages/sec whole-node, 1024x768 pages,
~40 crops/page, + p50 page latency).

  python bench.py --gpus N --steps K --warmup W

A *step* is `--reps` (1) passes of the hot path (resize/pad -> CRAFT -> union-find CCL -> calipers ->
crop-batch packer -> PARSeq -> token ids), each over one batch of `--pages` synthetic pages per GPU,
inputs already resident in HBM.  Page-level data parallelism: each rank owns its pages and a
full weights replica (weak scaling); for N > 1 the decoded token ids of every rank are
all-gathered with RCCL inside the timed region, device buffer to device buffer, by the engine's C++
host (include/tuatara_hip.h, "multi-GPU") - the only exchange the path has.  `python3 bench.py --gpus N`
starts its N ranks itself (tuatara_amd/launch.py: one child process per GPU, rank 0's line relayed, non-zero
exit when a rank fails or hangs); under `python -m torch.distributed.run ... bench.py --gpus N` the launcher's
ranks are used as they come.  The ranks never import torch.  Rank 0 prints ONE JSON line.

`roofline` follows SURVEY.md section 8(d): the dominant kernel (by time in the timed region), its launches and
average duration, `achieved` = its ALGORITHMIC flops / its time, `frac` = achieved / the dense f16 MFMA peak;
what the matrix pipe executes for it (x 3 or x 4 in the split-operand precision) is `mfma_pipe_frac`.

`value` is measured in the engine's DEFAULT precision, f16x4 (fp32-equivalent split-operand f16 MFMA: logits
within 1e-3 of the CPU fp32 reference, identical boxes and strings - tests/test_gpu_x4_parity.py); the bf16 and
fp32-MFMA engines are timed beside it on the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CRAFT_GFLOP_PER_PAGE = 559.5      # 27 convs, 2*MACs, BN folded, 1024x768
PARSEQ_GFLOP_PER_CROP = 6.129     # encoder 5.747 + KV-cached AR 0.190 + refine 0.191
MFMA_16BIT_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 / f16
HBM_PIN_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s
HBM_STREAM_GBPS = 6300.0          # what a streaming kernel reaches on this chip (the guide's 6.0 - 6.3 TB/s; profiles/r06_store_rate.txt: 6.2 - 6.8 for 16-byte stores of every shape)
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_PER_PRODUCT = {"f16x4": 4, "bf16": 1, "f32": 1}   # matrix-pipe flops per algorithmic flop (tuatara_amd/csrc/split.h): PARSeq
CRAFT_MFMA_PER_PRODUCT = {"f16x4": 3, "bf16": 1, "f32": 1}   # ... CRAFT runs on activation pairs: three MFMAs per product


# --------------------------------------------------------------------------------------------------------------- CPU baseline
def grid_rects(H2: int, W2: int, ratio: float):
    """The 5 x 8 grid of 150 x 40 px boxes the engine's `bench_grid_boxes` option puts on every page (heat-map units)."""
    import numpy as np
    out = []
    for r in range(8):
        for c in range(5):
            out.append([(c + 0.5) * W2 / 5.0, (r + 0.5) * H2 / 8.0, 75.0 * ratio, 20.0 * ratio, 0.0])
    return np.array(out, np.float32)


def cpu_baseline_child(args_json: str) -> None:
    """Runs in a FRESH process (no GPU runtime, no engine threads): the CPU path the reference runs (LibTorch fp32 + OpenCV),
    restated by the oracle (torch CPU fp32 + the C restatement of the OpenCV steps), on a bounded sample of the benchmark's
    pages and on the SAME 40-box grid the GPU leg recognises, under the two schedules of SURVEY.md section 8(d):

      best_effort         models loaded once, all crops of a page in one batch; torch threads swept, best kept
      reference_faithful  tuatara.cpp as written: both TorchScript archives loaded inside every call (:336, :428), the recogniser
                          in chunks of 4 crops (:452) on 6 threads sharing one module (:461-475)"""
    import queue
    import threading

    import numpy as np
    import torch

    from oracle import pipeline, post
    from tuatara_amd import synth
    from tuatara_amd import weights as W

    a = json.loads(args_json)
    H, Wd, words, grid = 1024, 768, a["words"], a["grid"]
    wdir = a["wdir"]
    os.makedirs(wdir, exist_ok=True)
    craft_state, parseq_state = W.synth_craft(0, True), W.synth_parseq(0)
    craft, parseq = pipeline.load_models(craft_state, parseq_state)
    pages = [synth.synthetic_page(sd, H, Wd, n_words=words, layout=a.get("layout", "jitter4")) for sd in range(a["n_pages"])]
    ncpu = os.cpu_count() or 1

    def one_page(img, det_model, rec):
        d = pipeline.detect(det_model, img)                                   # resize, CRAFT, CCL + boxes: always in full
        boxes = d["boxes"]
        if grid:
            H2, W2 = d["heat"].shape[:2]
            boxes = post.adjust_result_coordinates(grid_rects(H2, W2, float(d["ratio"])), 1.0 / float(d["ratio"]), 1.0 / float(d["ratio"]))
        crops = [c for c in (post.crop_resize(d["swapped"], b, True) for b in boxes) if c is not None]
        return rec(np.stack(crops)) if crops else []

    def rec_batch(crops):
        return post.decode_logits(pipeline.parseq_logits(parseq, crops, batch=len(crops)))[0]

    one_page(pages[0][:256, :256].copy(), craft, rec_batch)                    # warm-up (allocator, threads)
    sweep = []
    for nt in sorted({min(ncpu, t) for t in a["threads"]}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        ncrops = sum(len(one_page(pg, craft, rec_batch)) for pg in pages[:a["pages_per_setting"]])
        dt = time.perf_counter() - t0
        sweep.append({"torch_threads": nt, "pages": a["pages_per_setting"], "crops": ncrops, "seconds": dt, "pages_per_s": a["pages_per_setting"] / dt})
    best = dict(max(sweep, key=lambda s: s["pages_per_s"]))
    torch.set_num_threads(best["torch_threads"])
    # the reported figure: a longer run at the sweep's best setting (the sweep's two pages per setting only rank the settings)
    t0 = time.perf_counter()
    ncrops = sum(len(one_page(pg, craft, rec_batch)) for pg in pages[:a["pages_final"]])
    dt = time.perf_counter() - t0
    best.update({"pages": a["pages_final"], "crops": ncrops, "seconds": dt, "pages_per_s": a["pages_final"] / dt})
    cpath, ppath = os.path.join(wdir, "craft_traced_torchscript_model.pt"), os.path.join(wdir, "parseq_torchscript.bin")
    with torch.no_grad():
        canvas, _ = post.resize_aspect_ratio(np.ascontiguousarray(pages[0][:, :, ::-1]))
        torch.jit.trace(craft, torch.zeros(1, 3, canvas.shape[0], canvas.shape[1]), check_trace=False).save(cpath)
        torch.jit.trace(parseq, torch.zeros(4, 3, 32, 128), check_trace=False).save(ppath)

    def faithful(img):
        det = torch.jit.load(cpath)                                           # :333-336, per call
        rec = torch.jit.load(ppath)                                           # :423-428, per call

        def rec_chunks(crops):
            q, outs, lock = queue.Queue(), [], threading.Lock()
            for i in range(0, len(crops), 4):                                 # :450-459
                ch = crops[i:i + 4]
                n = len(ch)
                if n < 4:                                                     # the traced archive has a fixed batch of 4: pad the last chunk
                    ch = np.concatenate([ch, np.repeat(ch[-1:], 4 - n, 0)])
                q.put((i, n, torch.from_numpy(ch).permute(0, 3, 1, 2).float().div(255.0)))

            def infer():                                                      # :289-312
                while True:
                    try:
                        i, n, x = q.get_nowait()
                    except queue.Empty:
                        return
                    with torch.no_grad():
                        y = rec(x)[:n]
                    with lock:
                        outs.append((i, y))

            th = [threading.Thread(target=infer) for _ in range(6)]           # :461-475
            for t in th:
                t.start()
            for t in th:
                t.join()
            outs.sort(key=lambda t: t[0])                                     # :478
            return post.decode_logits(torch.cat([y for _, y in outs]).softmax(-1).numpy())[0]   # :485-486
        return one_page(img, det, rec_chunks)

    t0 = time.perf_counter()
    n_pf = a["pages_faithful"]
    n_f = sum(len(faithful(pg)) for pg in pages[:n_pf])
    dt_f = time.perf_counter() - t0
    print(json.dumps({
        "value": best["pages_per_s"], "unit": "pages/s", "cores": best["torch_threads"], "kind": "port", "nproc": ncpu,
        "sample": f"{best['pages']} of the benchmark's synthetic {H}x{Wd} pages at the best setting of a thread sweep ({a['pages_per_setting']} pages per setting), the same 40-box "
                  f"grid the GPU leg recognises ({best['crops']} crops), fresh process, models loaded once, one PARSeq batch per page, torch {torch.__version__} fp32",
        "thread_sweep": sweep,
        "schedules": {"best_effort": {"pages_per_s": best["pages_per_s"], "torch_threads": best["torch_threads"]},
                      "reference_faithful": {"pages_per_s": n_pf / dt_f, "pages": n_pf, "crops": n_f, "seconds": dt_f, "torch_threads": best["torch_threads"],
                                             "what": "TorchScript archives loaded per call, PARSeq in chunks of 4 on 6 threads (tuatara.cpp:336, :428, :452, :461)"}},
        "implementation": "Python port (oracle/): torch CPU fp32 + C restatement of the OpenCV steps"}))


def run_cpu_baseline(words: int, grid: int, layout: str) -> dict:
    a = {"words": words, "grid": grid, "layout": layout, "n_pages": 8, "pages_per_setting": 2, "pages_final": 8, "pages_faithful": 4, "threads": [8, 16, 32, 64, 128],
         "wdir": os.path.join(tempfile.gettempdir(), f"tuatara_bench_cpu_{os.getuid()}")}
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", json.dumps(a)], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=1500, check=True, text=True)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as ex:   # the baseline must never take the GPU number down with it
        err = getattr(ex, "stderr", "") or ""
        return {"value": None, "unit": "pages/s", "cores": None, "kind": "port", "sample": f"failed: {ex} {err[-300:]}"}


def shared_pages(seed_list, H, Wd, words, layout, rank, world):
    """seed -> page [H, Wd, 3] u8.  One rank alone renders its pages.  N ranks on one host all walk the same 512-seed stream (from different offsets): the set is
    rendered ONCE - by whichever rank takes the lock first - into a file under /dev/shm (or the temp directory), and every rank reads it; rank 0 removes the
    file when it has its copy and every rank has passed the lock.  (Eight ranks each drawing 512 pages with PIL was 8 x 12 s of host work before the first GPU call.)"""
    import fcntl

    import numpy as np

    from tuatara_amd import synth
    if world == 1:
        return {sd: synth.synthetic_page(sd, H, Wd, n_words=words, layout=layout) for sd in seed_list}
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    tag = f"tuatara_bench_pages_{os.getuid()}_{H}x{Wd}_{words}_{layout}_{len(seed_list)}_{seed_list[0]}_{seed_list[-1]}_{os.environ.get('MASTER_PORT', '0')}"
    path, lock = os.path.join(base, tag + ".u8"), os.path.join(base, tag + ".lock")
    n = len(seed_list)
    with open(lock, "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not (os.path.exists(path) and os.path.getsize(path) == n * H * Wd * 3):
                mm = np.memmap(path + ".tmp", np.uint8, "w+", shape=(n, H, Wd, 3))
                for k, sd in enumerate(seed_list):
                    mm[k] = synth.synthetic_page(sd, H, Wd, n_words=words, layout=layout)
                mm.flush()
                del mm
                os.replace(path + ".tmp", path)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    arr = np.fromfile(path, np.uint8).reshape(n, H, Wd, 3)        # this rank's own copy in memory
    return {sd: arr[k] for k, sd in enumerate(seed_list)}


def remove_shared_pages():
    import glob
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    for f in glob.glob(os.path.join(base, f"tuatara_bench_pages_{os.getuid()}_*_{os.environ.get('MASTER_PORT', '0')}.*")):
        try:
            os.remove(f)
        except OSError:
            pass


def stub_rank(kind: str, rank: int, world: int, wd, args) -> None:
    """A rank body without a GPU (TUATARA_BENCH_STUB, tests/test_launch_cpu.py): ok = every rank reports, rank 0 prints the line;
    fail = rank 1 leaves with status 7; hang = rank 1 never returns (the launcher's deadline ends it); stage = rank 1 sits in a watched
    stage past its allowance (the watchdog's status 3)."""
    if kind == "fail" and rank == 1:
        raise SystemExit(7)
    if kind == "hang" and rank == 1:
        time.sleep(3600)
    if kind == "stage" and rank == 1:
        with wd.stage("stub: communicator set-up", 0.5):
            time.sleep(3600)
    if kind in ("fail", "hang", "stage") and rank != 1:
        time.sleep(3600 if kind != "fail" else 30)      # the healthy ranks wait in their collective: the launcher must end them
    with wd.stage("stub: work", 30):
        time.sleep(0.2)
    if rank == 0:
        print("rank 0 chatter that is not the result line")
        print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": world, "steps": args.steps, "warmup": args.warmup}))


# --------------------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pages", type=int, default=64, help="pages per GPU per pass (PARSeq runs over all crops of a pass, in even groups of <= 1820 crops); 64 per pass "
                    "instead of round 4's 32: +1.3-1.5 %% pages/s on the same box (half as many hand-overs between the host and the two streams)")
    ap.add_argument("--reps", type=int, default=1, help="passes of --pages pages per step (a step carries >= 150 ms of GPU work)")
    ap.add_argument("--words", type=int, default=40, help="words drawn per synthetic page (SURVEY.md section 8d: ~40 random words)")
    ap.add_argument("--boxes", default="grid40", choices=["grid40", "detected"], help="grid40 (SURVEY.md section 8d): CRAFT + CCL + box extraction run in full (timed), then every page's "
                    "boxes are replaced by a fixed 5 x 8 grid of 150 x 40 px boxes so that PARSeq sees exactly 40 crops per page; detected: the synthetic detector's own boxes")
    ap.add_argument("--buffers", type=int, default=16, help="distinct device page buffers rotated over the passes (each holds --pages distinct pages of the 512-seed stream; 16 x 32 = all 512 seeds)")
    ap.add_argument("--precision", default="f16x4", choices=["f16x4", "bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the comparison legs after the timed region (bf16 / fp32 engines, detected boxes, latency, host buffers)")
    ap.add_argument("--latency-iters", type=int, default=20)
    ap.add_argument("--stream", type=int, default=1, help="1: feed the passes through ttr_stream_push (batch j's detector and batch j-1's recogniser are enqueued before batch j-2's results are awaited, "
                    "host box extraction overlaps GPU work; two streams by default: batch j-1's recogniser runs beside batch j's detector, tuning key recog_overlap); 0: one synchronous ttr_pages_to_data_dev call per pass")
    ap.add_argument("--tune", action="append", default=[], help="engine tuning knob key=value (ttr_engine_set_tuning), repeatable")
    ap.add_argument("--mode", default="throughput", choices=["throughput", "latency"], help="latency: single pages through the sharded path (rank 0 detects, the crop batch is "
                    "broadcast and recognised in shards, ids all-gathered): reports p50_page_latency_ms for N GPUs")
    ap.add_argument("--deadline", type=float, default=1500.0, help="N > 1 started from this one command: seconds the ranks get before the launcher ends them (exit status 124)")
    ap.add_argument("--stage-deadline", type=float, default=300.0, help="seconds a rank may spend in one of its watched stages (engine / communicator set-up, first gather) before it "
                    "leaves with status 3 naming the stage")
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child is not None:
        return cpu_baseline_child(args.cpu_baseline_child)

    from tuatara_amd import launch as L

    if L.wants_launch(args.gpus):
        # `python3 bench.py --gpus N` with no launcher around it: this process starts the N ranks itself (one per GPU, this same command line,
        # RANK / WORLD_SIZE / MASTER_* in their environment), relays rank 0's JSON line and leaves with their status; it never touches a GPU.
        # An external `python -m torch.distributed.run ... bench.py --gpus N` keeps working: its ranks arrive here with WORLD_SIZE set.
        status, _ = L.run_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, deadline_s=args.deadline)
        raise SystemExit(status)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    grid = 1 if args.boxes == "grid40" else 0
    # grid40: each word is drawn inside its cell's 150 x 40 px box, so the 40 crops of a page frame text (tuatara_amd/synth.py)
    layout = "cells5x8" if grid else "jitter4"
    wd = L.StageWatchdog(rank)            # a rank stuck in one stage leaves with status 3 and the stage's name
    stub = os.environ.get("TUATARA_BENCH_STUB")
    if stub:                              # tests/test_launch_cpu.py: the launcher and the watchdog with a rank body that needs no GPU
        return stub_rank(stub, rank, world, wd, args)

    # the CPU leg first, in a child process, while this process has not touched the GPU (and runs nothing else)
    # (N > 1: rank 0 runs it at the END instead, when every rank's GPU work is done and the other ranks have left - the host is then as quiet as here, and no
    # rank waits at a rendezvous for a CPU measurement)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "throughput":
        cpu = run_cpu_baseline(args.words, grid, layout)

    os.environ.setdefault("TUATARA_PRELOAD_TORCH", "0")   # the GPU processes of the benchmark stay torch-free (tuatara_amd/engine.py: load)
    import fcntl

    import numpy as np

    from tuatara_amd import build as B
    from tuatara_amd import synth
    from tuatara_amd import weights as W
    from tuatara_amd.engine import Comm, DeviceBuffer, Engine

    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    with open(os.path.join(ROOT, "build", ".bench_lock"), "w") as lk:      # one rank builds (a no-op when the library is current)
        fcntl.flock(lk, fcntl.LOCK_EX)
        B.build_lib()
        fcntl.flock(lk, fcntl.LOCK_UN)
    wdir = os.path.join(tempfile.gettempdir(), f"tuatara_bench_weights_{os.getuid()}_{local_rank}")
    W.make_synthetic_weights(wdir, seed=0, structured=True)

    P, H, Wd, R = args.pages, 1024, 768, max(1, args.reps)
    # TUATARA_BENCH_SHARE_GPU=1 (a pre-flight on a single-GPU box, never a benchmark): every rank on device 0, the exchange over the framed TCP
    # transport (RCCL refuses two ranks on one device) - launcher, ranks, header / counts / payload gathers and the result line, all but RCCL itself
    share = os.environ.get("TUATARA_BENCH_SHARE_GPU") == "1"
    with wd.stage("engine set-up (weights to the GPU)", args.stage_deadline):
        eng = Engine(wdir, precision=args.precision, device=0 if share else local_rank, bench_grid_boxes=grid)
    for kv in args.tune:
        k, v = kv.split("=")
        assert eng.set_tuning(k.encode(), int(v)) == 0, kv
    comm = None
    rank_map = None
    if world > 1:
        port = int(os.environ.get("TUATARA_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
        with wd.stage("communicator set-up (TCP rendezvous + ncclCommInitRank x 2)", args.stage_deadline):
            comm = Comm(eng, rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), port, transport="socket" if share else "rccl")
        with wd.stage("first host all-gather (barrier)", args.stage_deadline):
            comm.barrier()
        comm.attach(True)          # from here on every batch all-gathers its token ids on the engine's stream (ncclAllGather)
        rank_map = comm.describe_all()   # rank -> HIP device / PCI bus id / RCCL version / pid, on the result line

    def fence():
        if comm:
            comm.barrier()
        eng.lib.ttr_dev_sync(eng.h)
        if comm:
            comm.barrier()

    # ------------------------------------------------------------------ latency mode
    if args.mode == "latency":
        one = DeviceBuffer(H * Wd * 3)
        one.upload(synth.synthetic_page(0, H, Wd, n_words=args.words, layout=layout))
        if comm:
            comm.attach(False)
        call = (lambda: comm.pages_to_data_sharded(one if rank == 0 else None, 1, H, Wd)) if comm else (lambda: eng.pages_to_data_dev(one, 1, H, Wd))
        for _ in range(max(2, args.warmup)):
            call()
        lat = []
        fence()
        t0 = time.perf_counter()
        for _ in range(max(args.steps, 10)):
            t1 = time.perf_counter()
            res = call()
            lat.append((time.perf_counter() - t1) * 1e3)
        fence()
        dt = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"metric": "p50 page latency (1024x768 page, crop batch sharded over the GPUs)", "value": float(np.median(lat)), "unit": "ms",
                              "n_gpus": world, "steps": len(lat), "warmup": max(2, args.warmup), "ms_per_step": dt / len(lat) * 1e3, "higher_is_better": False,
                              "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                              "config": {"workload": "one synthetic 1024x768 page per call, latency mode (ttr_pages_to_data_dev_sharded)", "crops": len(res[0]) if res else 0,
                                         "boxes": args.boxes, "parallelism": f"crop-shard x{world}"},
                              "p50_page_latency_ms": float(np.median(lat)), "p90_page_latency_ms": float(np.quantile(lat, 0.9))}))
        if comm:
            comm.barrier()
            comm.close()
        return

    # ------------------------------------------------------------------ throughput mode
    # the 512-seed stream of SURVEY.md section 8d: pass k of rank r works on pages (r * NB + k % NB) * P .. + P - 1 (mod 512), NB distinct
    # device-resident buffers rotated so that consecutive passes never see the same pages
    NB = max(3, args.buffers)
    seeds = [[(((rank * NB + b) * P + i) % 512) for i in range(P)] for b in range(NB)]
    with wd.stage("rendering the synthetic pages", args.stage_deadline):
        page_of_seed = shared_pages(sorted({sd for b in seeds for sd in b}), H, Wd, args.words, layout, rank, world)
    host_pages = [[page_of_seed[sd] for sd in seeds[b]] for b in range(NB)]
    dbufs = []
    for b in range(NB):
        d = DeviceBuffer(P * H * Wd * 3)
        d.upload(np.stack(host_pages[b]))
        dbufs.append(d)
    pass_no = [0]
    stream = bool(args.stream)

    def run_passes(e, k, keep=None):
        """k passes of P pages through engine e; every pass completed when this returns.  keep: list that receives the results."""
        for _ in range(k):
            bsel = pass_no[0] % NB                # a buffer is pushed again NB >= 3 pushes later: its results came back before
            pass_no[0] += 1
            prev = e.stream_push(dbufs[bsel], P, H, Wd) if stream else e.pages_to_data_dev(dbufs[bsel], P, H, Wd)
            if prev and keep is not None:
                keep.append(prev)
        while stream:                            # the (up to two) batches still in flight
            last = e.stream_flush()
            if not last:
                break
            if keep is not None:
                keep.append(last)

    with wd.stage("first pass (workspaces, first token-id all-gather)", args.stage_deadline):
        run_passes(eng, 1)         # untimed, before the warm-up proper: the one pass that allocates, and with N > 1 the first gather
    if world > 1 and rank == 0:
        remove_shared_pages()      # (every rank has its copy: they all took part in the gather above)
    if args.warmup:
        with wd.stage("warm-up passes", args.stage_deadline + 30.0 * args.warmup * R):
            run_passes(eng, args.warmup * R)
    eng.set_profiling(1)           # timed region: HIP events around the dominant kernels only (the CRAFT convolutions)
    fence()
    t0 = time.perf_counter()
    kept = []
    run_passes(eng, args.steps * R, kept)
    fence()
    dt = time.perf_counter() - t0
    prof = eng.get_profile()
    kinds = eng.get_profile_kinds()
    eng.set_profiling(0)
    stage = eng.last_stage_ms()
    # In the timed region a batch's recogniser runs on its own stream beside the next batch's detector (tuning key "recog_overlap", the engine's default: +3 % pages/s):
    # the dominant kernel's launches then share the chip with the recogniser's, and their durations say so.  The same kernel ALONE on the chip: a few passes
    # with the overlap off, outside the timed region -> roofline.*_exclusive
    kinds_excl = None
    overlap_on = not any(kv.startswith("recog_overlap=0") for kv in args.tune) and stream
    if overlap_on and not args.no_extras and world == 1:
        eng.set_tuning(b"recog_overlap", 0)
        run_passes(eng, 1)
        eng.lib.ttr_dev_sync(eng.h)
        eng.get_profile_kinds()
        eng.set_profiling(1)
        before = {(k["kind"], k["stage"]): dict(k) for k in eng.get_profile_kinds()}
        run_passes(eng, 4)
        eng.lib.ttr_dev_sync(eng.h)
        after = eng.get_profile_kinds()
        eng.set_profiling(0)
        eng.set_tuning(b"recog_overlap", 1)
        kinds_excl = []
        for k in after:
            b = before.get((k["kind"], k["stage"]), {"ms": 0.0, "launches": 0, "alg_flops": 0.0, "exec_flops": 0.0})
            kinds_excl.append({"kind": k["kind"], "stage": k["stage"], "ms": k["ms"] - b["ms"], "launches": k["launches"] - b["launches"],
                               "alg_flops": k["alg_flops"] - b["alg_flops"], "exec_flops": k["exec_flops"] - b["exec_flops"]})
    gathered_rows = None
    if comm:
        cts, gids = comm.last_gathered()
        gathered_rows = int(len(gids))
        assert cts.shape == (world, P) and len(gids) == int(cts.sum()), (cts.shape, len(gids))
        if grid:                                   # the fixed grid: every page of every rank contributes exactly 40 rows of 26 ids
            assert (cts == 40).all() and gathered_rows == world * P * 40, (gathered_rows, world, P)
        assert gids.shape[1] == 26 and (gids >= 0).all() and (gids < 98).all()
        dts = comm.allgather_host(np.array([dt], np.float64)).reshape(-1)     # every rank's own time for its K steps
        rank_rates = [float(P * args.steps * R / t) for t in dts]               # pages/s of each rank (weak scaling: equal work per rank)
        dt = float(dts.max())                                                    # MAX over ranks
    if not comm:
        rank_rates = [float(P * args.steps * R / dt)]
    res = kept[-1]
    crops_per_page = float(np.mean([len(r) for r in res]))
    lens = np.bincount([len(t) for batch in kept for r in batch for t in r.texts], minlength=27)

    out = None
    n_pass = args.steps * R
    if rank == 0:
        peak = MFMA_F32_PEAK_TFLOPS if args.precision == "f32" else MFMA_16BIT_PEAK_TFLOPS
        craft_kinds = [k for k in kinds if k["stage"] == 0 and k["ms"] > 0]

        def line(k):
            """one kernel kind -> its roofline figures: ALGORITHMIC flops / time is `achieved`; what the matrix pipe executes beside it"""
            sec = k["ms"] * 1e-3
            d = {"kernel": k["kind"], "launches_per_pass": k["launches"] / max(1, n_pass), "avg_launch_us": k["ms"] * 1e3 / max(1, k["launches"]),
                 "algorithmic_gflop_per_launch": k["alg_flops"] / max(1, k["launches"]) / 1e9,
                 "achieved": k["alg_flops"] / sec / 1e12, "frac": k["alg_flops"] / sec / 1e12 / peak,
                 "mfma_pipe_tflops": k["exec_flops"] / sec / 1e12, "mfma_pipe_frac": k["exec_flops"] / sec / 1e12 / peak,
                 "mfma_flops_per_algorithmic_flop": k["exec_flops"] / k["alg_flops"] if k["alg_flops"] else None}
            # which roof binds this kind, and how far it is: the matrix pipe (executed flops / dense f16 peak) against the memory (ALGORITHMIC bytes - every operand
            # read once, every result written once, at the engine's plane sizes - / time, over the 6.3 TB/s a streaming kernel reaches on this chip,
            # profiles/r06_store_rate.txt and MI355X_MICROARCH.md; the 8 TB/s pin rate beside it)
            by = k.get("alg_bytes") or 0.0
            if by > 0:
                gbps = by / sec / 1e9
                d.update({"algorithmic_MB_per_launch": by / max(1, k["launches"]) / 1e6, "hbm_GBps": gbps, "hbm_frac_of_6300": gbps / HBM_STREAM_GBPS, "hbm_frac_of_8000": gbps / HBM_PIN_GBPS})
                d["bound"] = "hbm" if gbps / HBM_STREAM_GBPS > d["mfma_pipe_frac"] else "mfma"
                d["frac_of_binding_roof"] = max(gbps / HBM_STREAM_GBPS, d["mfma_pipe_frac"])
            else:
                d["bound"] = "mfma"
                d["frac_of_binding_roof"] = d["mfma_pipe_frac"]
            return d

        dom = max(craft_kinds, key=lambda k: k["ms"]) if craft_kinds else None
        fam_ms = sum(k["ms"] for k in craft_kinds)
        fam_alg, fam_exec = sum(k["alg_flops"] for k in craft_kinds), sum(k["exec_flops"] for k in craft_kinds)
        # HBM bytes per launch of the dominant kernel: the two --pmc passes (FETCH_SIZE doubled, WRITE_SIZE) committed under profiles/
        traffic = traffic_src = fam_traffic = det_gb_page = None
        for name in (("r06_pmc_craft_x4.json", "r05_pmc_craft_x4.json", "r04_pmc_craft_x4.json", "r03_pmc_craft_x4.json") if args.precision == "f16x4" else ("r02_pmc_craft_b16_v2.json",)):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    tj = json.load(f)
                fam_traffic = tj["craft_conv_kernels"]["hbm_bytes_per_launch"]
                det_gb_page = (tj.get("detector_all_kernels") or {}).get("hbm_GB_per_page")
                tag = dom["kind"].split("<")[0] if dom else ""
                width = dom["kind"].split("<")[1].split(",")[0] if dom and "<" in dom["kind"] else ""
                for kn, kv in tj["kernels"].items():
                    if tag and tag in kn and (f"<{width}," in kn or f"ILi{width}E" in kn):
                        traffic = kv["fetch_bytes_per_launch"] + kv["write_bytes_per_launch"]
                traffic_src = {"file": "profiles/" + name, "build": tj.get("build"), "pages_per_group": tj.get("pages")}
                break
            except Exception:
                pass
        total_pages = world * P * n_pass
        roof = {"bound": "mfma", "peak": peak, "unit": "TFLOP/s"}
        if dom:
            roof.update(line(dom))
            roof["achieved_is"] = ("ALGORITHMIC flops of this kernel's launches (2 x MACs of the unpadded layers it runs: SURVEY.md section 8(d)) / their summed durations, "
                                   "HIP events on the engine's stream inside the timed region; a run of consecutive launches of the kernel shares one event pair, so the "
                                   "dispatch gaps inside a run are charged to it.  mfma_pipe_* = the flops the matrix cores execute for them (three f16 MFMAs per product "
                                   "on activation pairs, four on exact triples: tuatara_amd/csrc/split.h)")
            roof["share_of_craft_time"] = dom["ms"] / fam_ms if fam_ms else None
            if overlap_on:
                roof["shares_the_chip"] = ("in the timed region this kernel's launches run while the previous batch's recogniser is on the chip too (engine default \"recog_overlap\": "
                                           "the recogniser on its own stream, +3 % pages/s): avg_launch_us / achieved / frac are what the kernel gets of a shared machine; "
                                           "the *_exclusive keys = the same launches alone on the chip (overlap off, passes behind the timed region)")
            if kinds_excl:
                ke = [k for k in kinds_excl if k["kind"] == dom["kind"] and k["stage"] == 0 and k["ms"] > 0 and k["launches"] > 0]
                if ke:
                    e = line(ke[0])
                    e["launches_per_pass"] = ke[0]["launches"] / 4.0          # (four passes were measured)
                    # (scalar keys: a nested object did not survive into the driver's parsed record in round 5)
                    roof.update({"frac_exclusive": e["frac"], "achieved_exclusive": e["achieved"], "avg_launch_us_exclusive": e["avg_launch_us"],
                                 "mfma_pipe_frac_exclusive": e["mfma_pipe_frac"], "mfma_pipe_tflops_exclusive": e["mfma_pipe_tflops"], "launches_per_pass_exclusive": e["launches_per_pass"]})
        try:
            with open(os.path.join(ROOT, ".build_hash")) as f:
                build_now = f.read().strip()
        except OSError:
            build_now = None
        roof.update({"traffic": traffic, "traffic_unit": "HBM bytes per launch of this kernel (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes over one CRAFT group)",
                     "traffic_profile": (traffic_src or {}).get("file"), "traffic_profile_build": (traffic_src or {}).get("build"), "build_hash": build_now,
                     "traffic_note": "--pmc cannot run inside this process: the counters come from the committed profile named here, taken on the build traffic_profile_build; "
                                     "build_hash is the library this line was measured on"})
        out = {
            "metric": "pages/sec whole-node (1024x768, ~40 crops/page)", "value": total_pages / dt, "unit": "pages/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("configs[4]: synthetic stream of 1024x768 pages, exactly 40 crops each (CRAFT + CCL + box extraction in full, then a fixed 5 x 8 grid "
                                    "of boxes goes to the recogniser: --boxes=grid40), page-level DP, RCCL all-gather of token ids in the C++ host") if grid else
                                   "configs[4]: synthetic stream of 1024x768 pages, the synthetic detector's own boxes, page-level DP, RCCL all-gather of token ids in the C++ host",
                       "pages_per_gpu_per_pass": P, "passes_per_step": R, "pages_per_gpu_per_step": P * R, "ms_per_pass": dt / n_pass * 1e3, "batches_in_flight": 3 if stream else 1,
                       "words_drawn_per_page": args.words, "page_layout": layout + (" (one word inside each of the 40 grid boxes)" if grid else ""), "crops_per_page": round(crops_per_page, 1), "boxes": args.boxes,
                       "ar_steps": "early exit when every crop of the batch has emitted EOS (upstream PARSeq's break, system.py); value_full_ar runs all 26 steps",
                       "decoded_string_length_histogram": lens.tolist(),
                       "distinct_pages": len({sd for b in seeds for sd in b}), "page_buffers_rotated": NB, "weights": "seeded synthetic (designed read-outs on random CRAFT / PARSeq, tuatara_amd/weights.py)",
                       "precision": {"f16x4": "fp32-equivalent split-operand f16 MFMA (tuatara_amd/csrc/split.h): PARSeq on exact activation triples x weight pairs (four MFMAs per product), "
                                              "CRAFT on activation pairs (three; its heat map stays at fp32 noise level); logits within 1e-3 of the CPU fp32 reference, boxes and "
                                              "strings identical (tests/test_gpu_x4_parity.py)",
                                     "bf16": "operands rounded to bf16: NOT output-equivalent (|dlogit| up to ~1e-1..1)", "f32": "fp32 MFMA"}[args.precision],
                       "parallelism": f"dp{world}" + (" (ranks SHARE one GPU over the TCP transport: a pre-flight, not a measurement)" if share else "")},
            "gathered_id_rows_last_pass": gathered_rows,
            "per_rank_pages_per_s": {"min": min(rank_rates), "max": max(rank_rates), "by_rank": [round(x, 2) for x in rank_rates],
                                     "is": "each rank's own pages/s over the timed steps (value = all ranks' pages / the slowest rank's time)"},
            "ranks": rank_map,
            "stage_ms_last_pass": {k: round(v, 3) for k, v in stage.items()},
            "roofline": roof,
            "roofline_craft_family": {"kernel": "all CRAFT convolution launches", "bound": "mfma", "peak": peak, "unit": "TFLOP/s",
                                      "achieved": fam_alg / (fam_ms * 1e-3) / 1e12 if fam_ms else None, "frac": fam_alg / (fam_ms * 1e-3) / 1e12 / peak if fam_ms else None,
                                      "mfma_pipe_tflops": fam_exec / (fam_ms * 1e-3) / 1e12 if fam_ms else None,
                                      "mfma_pipe_frac": fam_exec / (fam_ms * 1e-3) / 1e12 / peak if fam_ms else None,
                                      "ms_per_pass": fam_ms / max(1, n_pass), "algorithmic_gflop_per_page": fam_alg / max(1, P * n_pass) / 1e9,
                                      "survey_gflop_per_page": CRAFT_GFLOP_PER_PAGE, "traffic_bytes_per_launch_all_kinds": fam_traffic,
                                      "hbm_GB_per_page_every_detector_kernel": det_gb_page,   # (convolutions + pools + resize: the same two --pmc passes)
                                      "by_kernel": sorted((line(k) for k in craft_kinds), key=lambda d: -d["avg_launch_us"] * d["launches_per_pass"])},
        }

    # ------------------------------------------------------------------ comparison legs (outside the timed region; 1 GPU only)
    if world == 1 and not args.no_extras:
        def rate(e, passes):
            run_passes(e, 1)
            e.lib.ttr_dev_sync(e.h)
            t1 = time.perf_counter()
            kk = []
            run_passes(e, passes, kk)
            e.lib.ttr_dev_sync(e.h)
            return P * passes / (time.perf_counter() - t1), float(np.mean([len(r) for r in kk[-1]]))

        # secondary rooflines (ViT / decoder GEMMs): one pass with every launch bracketed by events, by kernel kind
        eng.set_profiling(2)
        run_passes(eng, 1)
        eng.lib.ttr_dev_sync(eng.h)
        allk = eng.get_profile_kinds()
        pk = [k for k in allk if k["stage"] == 1 and k["ms"] > 0]
        ck = [k for k in allk if k["stage"] == 0 and k["ms"] > 0 and " | " in k["kind"]]      # the detector's launches of that pass, one kind per LAYER (engine_craft.cpp)
        eng.set_profiling(0)
        if ck:
            def lline(k):
                d = line(k)
                d["layer"], d["kernel"] = k["kind"].split(" | ", 1)
                d["launches_per_pass"] = k["launches"]
                return d
            out["roofline_craft_layers"] = {
                "is": "every detector layer's own roof (VERDICT r05 next 8): one extra pass behind the timed region with EVERY launch bracketed by HIP events on the engine's stream "
                      "(a lone batch: the detector runs without a recogniser beside it); bound = \"mfma\" (executed flops / dense f16 peak) or \"hbm\" (algorithmic bytes / time over the "
                      "6.3 TB/s a streaming kernel reaches on this chip), frac_of_binding_roof = how far that roof is",
                "by_layer": [lline(k) for k in ck]}
        peak = MFMA_F32_PEAK_TFLOPS if args.precision == "f32" else MFMA_16BIT_PEAK_TFLOPS

        def pline(k):
            sec = k["ms"] * 1e-3
            return {"kernel": k["kind"], "launches_per_pass": k["launches"], "avg_launch_us": k["ms"] * 1e3 / max(1, k["launches"]),
                    "algorithmic_gflop_per_launch": k["alg_flops"] / max(1, k["launches"]) / 1e9, "achieved": k["alg_flops"] / sec / 1e12,
                    "frac": k["alg_flops"] / sec / 1e12 / peak, "mfma_pipe_frac": k["exec_flops"] / sec / 1e12 / peak}

        p_ms, p_alg, p_exec = sum(k["ms"] for k in pk), sum(k["alg_flops"] for k in pk), sum(k["exec_flops"] for k in pk)
        out["roofline_parseq_gemm"] = {"kernel": "PARSeq batched matrix launches (encoder linears, the fused qkv + attention launch, cross K/V, refinement pass)", "bound": "mfma",
                                       "peak": peak, "unit": "TFLOP/s", "achieved": p_alg / (p_ms * 1e-3) / 1e12 if p_ms else None,
                                       "frac": p_alg / (p_ms * 1e-3) / 1e12 / peak if p_ms else None,
                                       "mfma_pipe_tflops": p_exec / (p_ms * 1e-3) / 1e12 if p_ms else None, "mfma_pipe_frac": p_exec / (p_ms * 1e-3) / 1e12 / peak if p_ms else None,
                                       "achieved_is": "per-launch algorithmic flops (2 x MACs of each layer; attention: 4 x keys x d per query and head) / per-launch durations",
                                       "launches_per_pass": sum(k["launches"] for k in pk), "ms_per_pass": p_ms, "measured": "one extra pass after the timed region, every launch bracketed",
                                       "by_kernel": sorted((pline(k) for k in pk), key=lambda d: -d["avg_launch_us"] * d["launches_per_pass"])}
        # all 26 AR steps (no early exit from the autoregressive loop)
        eng.set_tuning(b"ar_early_exit", 0)
        out["value_full_ar"], _ = rate(eng, 2)
        eng.set_tuning(b"ar_early_exit", 1)
        # detected boxes instead of the grid
        eng.set_tuning(b"bench_grid_boxes", 0 if grid else 1)
        run_passes(eng, 2)                       # (the crop count changes from pass to pass: let the workspaces grow first)
        r_det, cpp_det = rate(eng, 3)
        eng.set_tuning(b"bench_grid_boxes", grid)
        out["value_detected_boxes" if grid else "value_grid40"] = r_det
        out["crops_per_page_detected_boxes" if grid else "crops_per_page_grid40"] = cpp_det
        # p50 single-page latency (one page per call, synchronous)
        lat = []
        one = DeviceBuffer(H * Wd * 3)
        one.upload(host_pages[0][0])
        for _ in range(args.latency_iters):
            t1 = time.perf_counter()
            eng.pages_to_data_dev(one, 1, H, Wd)
            lat.append((time.perf_counter() - t1) * 1e3)
        out["p50_page_latency_ms"] = float(np.median(lat)) if lat else None
        hu = eng.last_host_us()          # host wall-clock splits of the last single-page call (microseconds)
        out["single_page_host_us"] = {"enqueue_detector": hu[0], "wait_for_ccl": hu[1], "copy_components": hu[2], "calipers_and_boxes": hu[3],
                                      "enqueue_recogniser": hu[4], "wait_for_ids": hu[5], "decode_strings": hu[7]}
        # the same workload from HOST memory through the drop-in surface's list form (ttr_images_to_data = pytuatara.images_to_data): 4 passes' worth of numpy pages
        # (pageable), bucketed, gathered into pinned staging by a helper thread and copied on an upload stream while the previous batch is on the GPU - what a
        # caller holding host buffers pays, copy included -- never `value`
        def host_call(nb):
            lst = [pg for b in range(nb) for pg in host_pages[b % len(host_pages)]]
            eng.lib.ttr_dev_sync(eng.h)
            t1 = time.perf_counter()
            eng.images_to_data(lst, keep=False)
            return len(lst), time.perf_counter() - t1
        host_call(1)                                           # (staging buffers are allocated on first use)
        n_a, t_a = host_call(4)
        n_b, t_b = host_call(12)
        # one call pays the pipeline's fill and drain once (the first detector and the last recogniser run alone); the marginal rate between a 4-batch and a
        # 12-batch call is the steady state a long list sees, and the figure to hold against `value`
        out["h2d_included_pages_per_s"] = (n_b - n_a) / (t_b - t_a)
        out["h2d_included_one_call"] = {"pages": n_b, "pages_per_s": n_b / t_b, "pages_short": n_a, "pages_per_s_short": n_a / t_a}
        out["h2d_included_is"] = ("ttr_images_to_data (= pytuatara.images_to_data) over host numpy pages, results returned to the caller: marginal rate between a "
                                  f"{n_a}-page and a {n_b}-page call; h2d_included_one_call holds the whole-call rates (pipeline fill and drain included)")
        # the other precisions on the same workload
        if args.precision == "f16x4":
            eb = Engine(wdir, precision="bf16", device=local_rank, bench_grid_boxes=grid)
            latb = []
            for _ in range(args.latency_iters):
                t1 = time.perf_counter()
                eb.pages_to_data_dev(one, 1, H, Wd)
                latb.append((time.perf_counter() - t1) * 1e3)
            out["bf16_p50_page_latency_ms"] = float(np.median(latb)) if latb else None
            out["bf16_pages_per_s"], _ = rate(eb, 6)
            eb.set_tuning(b"ar_early_exit", 0)
            out["bf16_full_ar_pages_per_s"], _ = rate(eb, 4)
            out["bf16_note"] = ("bf16 operands: NOT output-equivalent to the fp32 reference (tests/test_gpu_bf16_parity.py: margin rule only); bf16_pages_per_s uses upstream "
                                "PARSeq's early exit from the AR loop (the synthetic strings end within 10 characters), bf16_full_ar_pages_per_s runs all 26 steps")
            eb.close()
            ef = Engine(wdir, precision="f32", device=local_rank, bench_grid_boxes=grid)
            out["f32_mfma_pages_per_s"], _ = rate(ef, 1)
            ef.close()
    if comm:
        comm.barrier()
        comm.close()
        comm = None
    if rank == 0:
        if world > 1 and not args.no_cpu_baseline:
            eng.close()                   # the GPU is released; the other ranks are leaving or gone
            time.sleep(2.0)
            cpu = run_cpu_baseline(args.words, grid, layout)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))


if __name__ == "__main__":
    main()
