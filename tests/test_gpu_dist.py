"""-m gpu: the multi-GPU entry points of the C ABI (include/tuatara_hip.h, "multi-GPU") on the one MI355X of a GPU box: world_size 1
through the real RCCL calls (ncclCommInitRank, ncclAllGather on the engine's stream, ncclBroadcast) - a page with more than 128 crops
goes through the gather untruncated, latency mode returns what the plain call returns - once inside the test process (where torch
is imported: the engine then shares torch's bundled ROCm runtime and RCCL) and once in a fresh torch-free interpreter (the system
runtime, the way bench.py runs).  world_size 2 runs over gloo in tests/test_dist_cpu.py; N = 2, 4, 8 in the driver's scaling bench."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

BODY = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
from tuatara_amd import synth
from tuatara_amd.engine import Comm, DeviceBuffer, Engine
assert "torch" not in sys.modules
eng = Engine({wdir!r}, canvas_size=2048)
page = synth.synthetic_page(7, 2048, 1536, n_words=160)                    # 40 rows of 4 words on a 2048 canvas: > 128 boxes
small = [synth.synthetic_page(8 + i, 1024, 768, n_words=12) for i in range(3)]
comm = Comm(eng, 0, 1, unique_id=Comm.unique_id())
ref = eng.image_to_data(page)
assert len(ref) > 128, len(ref)
comm.attach(True)
buf = DeviceBuffer(page.nbytes); buf.upload(page)
res = eng.pages_to_data_dev(buf, 1, 2048, 1536)
counts, ids = comm.last_gathered()
assert counts.tolist() == [[len(ref)]] and ids.shape == (len(ref), 26)
assert np.array_equal(ids, np.array([r["ids"] for r in ref])) and [r["text"] for r in res[0]] == [r["text"] for r in ref]
b3 = DeviceBuffer(3 * 1024 * 768 * 3); b3.upload(np.stack(small))
got = []
for k in range(3):                                                           # streamed batches: the gather rides every pass
    prev = eng.stream_push(b3, 3, 1024, 768)
    if prev:
        c, i = comm.last_gathered(); got.append((prev, c, i))
while True:
    last = eng.stream_flush()
    if not last:
        break
    c, i = comm.last_gathered(); got.append((last, c, i))
assert len(got) == 3
for r, c, i in got:
    assert c.tolist() == [[len(p) for p in r]] and np.array_equal(i, np.concatenate([p.ids for p in r]))
comm.attach(False)
one = DeviceBuffer(small[0].nbytes); one.upload(small[0])
lat = comm.pages_to_data_sharded(one, 1, 1024, 768)                           # latency mode, world 1: detect, broadcast, recognise, gather
plain = eng.image_to_data(small[0])
assert len(plain) > 5 and [x["text"] for x in lat[0]] == [x["text"] for x in plain] and [x["bbox"] for x in lat[0]] == [x["bbox"] for x in plain]
assert comm.allgather_host(np.arange(5, dtype=np.int32)).tolist() == [[0, 1, 2, 3, 4]]
info = comm.describe_all()                                                     # what a scaling run prints: rank -> device / bus id / RCCL version
assert len(info) == 1 and info[0]["rank"] == 0 and info[0]["world"] == 1 and info[0]["transport"] == "rccl" and info[0]["rccl_version"] > 20000, info
assert info[0]["device"] == 0 and info[0]["pci_bus_id"] and info[0]["gpu"].startswith("gfx950") and info[0]["pid"] == os.getpid(), info
comm.close()
after = eng.image_to_data(small[0])
assert [x["text"] for x in after] == [x["text"] for x in plain]
print("OK", len(ref))
'''


def test_rccl_gather_and_latency_mode_in_a_torch_free_process(weights):
    code = BODY.format(root=ROOT, wdir=weights["dir"])
    env = dict(os.environ, TUATARA_PRELOAD_TORCH="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_rccl_gather_beside_torch(weights):
    """The same calls inside this process (torch imported by the oracle fixtures)."""
    from tuatara_amd import synth
    from tuatara_amd.engine import Comm, DeviceBuffer
    from tests.conftest import _engine
    eng = _engine(weights["dir"], "f16x4")
    comm = Comm(eng, 0, 1, unique_id=Comm.unique_id())
    pages = np.stack([synth.synthetic_page(20 + i, 1024, 768, n_words=10 + 5 * i) for i in range(2)])
    buf = DeviceBuffer(pages.nbytes)
    buf.upload(pages)
    comm.attach(True)
    res = eng.pages_to_data_dev(buf, 2, 1024, 768)
    counts, ids = comm.last_gathered()
    assert counts.tolist() == [[len(res[0]), len(res[1])]] and len(res[0]) != len(res[1])
    assert np.array_equal(ids, np.concatenate([res[0].ids, res[1].ids]))
    comm.attach(False)
    lat = comm.pages_to_data_sharded(buf, 2, 1024, 768)
    assert [[x["text"] for x in p] for p in lat] == [[x["text"] for x in p] for p in res]
    comm.close()
    buf.free()


# ------------------------------------------------------------------------------------------------------------ world size 2 on ONE GPU
# RCCL refuses two ranks on one device, so the two processes speak through the engine's socket transport (ttr_comm_create_socket): the
# SAME C++ code paths as over RCCL - header exchange, counts then payload, the streamed batches' gather on every pass, latency mode's
# broadcast + shard + gather, failure propagation - only the three collective calls themselves go over TCP (and are checked call by call).
RANK2 = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, {root!r})
from tuatara_amd import synth
from tuatara_amd.engine import Comm, DeviceBuffer, Engine, EngineError
rank, world, port = int(sys.argv[1]), 2, int(sys.argv[2])
eng = Engine({wdir!r})
comm = Comm(eng, rank, world, "127.0.0.1", port, transport="socket")
assert comm.lib.ttr_comm_transport(comm.h) == b"socket"
# every rank its own pages (page-level DP): rank r takes seeds 40 + 3 r ..., with different word counts so that the totals differ
mine = np.stack([synth.synthetic_page(40 + 3 * rank + i, 1024, 768, n_words=6 + 7 * rank + 3 * i) for i in range(3)])
other = np.stack([synth.synthetic_page(40 + 3 * (1 - rank) + i, 1024, 768, n_words=6 + 7 * (1 - rank) + 3 * i) for i in range(3)])
buf = DeviceBuffer(mine.nbytes); buf.upload(mine)
obuf = DeviceBuffer(other.nbytes); obuf.upload(other)
solo_mine = eng.pages_to_data_dev(buf, 3, 1024, 768)                 # no communicator attached yet: plain calls
solo_other = eng.pages_to_data_dev(obuf, 3, 1024, 768)
by_rank = [solo_mine, solo_other] if rank == 0 else [solo_other, solo_mine]
want_counts = [[len(p) for p in by_rank[r]] for r in range(world)]
want_ids = np.concatenate([p.ids for r in range(world) for p in by_rank[r]])
assert sum(want_counts[0]) != sum(want_counts[1])                     # ragged totals: the payload is padded to the larger one
comm.attach(True)
res = eng.pages_to_data_dev(buf, 3, 1024, 768)                        # synchronous batch: header, counts, payload
c, ids = comm.last_gathered()
assert c.tolist() == want_counts, (c.tolist(), want_counts)
assert np.array_equal(ids, want_ids)
assert [[x["text"] for x in p] for p in res] == [[x["text"] for x in p] for p in solo_mine]
got = []
for k in range(4):                                                    # streamed batches, three in flight: the gather rides every pass
    prev = eng.stream_push(buf, 3, 1024, 768)
    if prev:
        got.append(comm.last_gathered())
while True:
    last = eng.stream_flush()
    if not last:
        break
    got.append(comm.last_gathered())
assert len(got) == 4
for c, ids in got:
    assert c.tolist() == want_counts and np.array_equal(ids, want_ids)
# a rank that passes another page count: the call fails on BOTH ranks (status / page header), nobody is left in a gather
try:
    eng.pages_to_data_dev(buf, 3 if rank == 0 else 2, 1024, 768)
    raise SystemExit("mismatched page counts went through")
except EngineError as ex:
    assert "pages" in str(ex), str(ex)
# a rank whose detector fails (rank 1: an image too thin to resize): both ranks raise, the next batch works again
thin = DeviceBuffer(3 * 4 * 3000 * 3)
try:
    if rank == 1:
        eng.pages_to_data_dev(thin, 3, 1, 3000)
    else:
        eng.pages_to_data_dev(buf, 3, 1024, 768)
    raise SystemExit("a failed rank went unnoticed")
except EngineError as ex:
    assert ("rank 1 failed" in str(ex)) if rank == 0 else ("thin" in str(ex) or "resize" in str(ex)), str(ex)
res = eng.pages_to_data_dev(buf, 3, 1024, 768)
c, ids = comm.last_gathered()
assert c.tolist() == want_counts and np.array_equal(ids, want_ids)
comm.attach(False)
# latency mode: rank 0 detects and packs, the crop batch is broadcast, each rank recognises its shard, the ids are gathered
lat = comm.pages_to_data_sharded(obuf if rank == 0 else None, 3 if rank == 0 else 0, 1024, 768)
if rank == 0:
    assert [[x["text"] for x in p] for p in lat] == [[x["text"] for x in p] for p in solo_other]
    assert [[x["bbox"] for x in p] for p in lat] == [[x["bbox"] for x in p] for p in solo_other]
assert comm.allgather_host(np.array([10 + rank], np.int32)).ravel().tolist() == [10, 11]
comm.close()
print("OK rank", rank, json.dumps(want_counts))
'''


def test_world_size_two_on_one_gpu_over_the_socket_transport(weights):
    """Two torch-free processes, one engine each, the same GPU: throughput mode (synchronous and streamed), both failure modes, latency mode."""
    from tuatara_amd.launch import free_port
    port = free_port()
    code = RANK2.format(root=ROOT, wdir=weights["dir"])
    env = dict(os.environ, TUATARA_PRELOAD_TORCH="0", TUATARA_COMM_TIMEOUT="240")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=900)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and f"OK rank {r}" in o, (r, rc, o[-800:], e[-3000:])


def test_bench_starts_its_own_ranks_two_ranks_on_one_gpu(weights):
    """`python3 bench.py --gpus 2` as the driver would start it, on a single-GPU box: the launcher's two ranks share device 0 and speak over the
    framed TCP transport (TUATARA_BENCH_SHARE_GPU=1: a pre-flight, not a measurement) - launcher, rank bodies, stage watchdog, header / counts /
    payload gathers of every streamed pass, the max-over-ranks time and the single JSON line, all but RCCL itself."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(TUATARA_BENCH_SHARE_GPU="1", TUATARA_PRELOAD_TORCH="0", TUATARA_COMM_TIMEOUT="240")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pages", "8", "--buffers", "3", "--deadline", "800"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["gathered_id_rows_last_pass"] == 2 * 8 * 40                      # every page of both ranks: 40 rows of 26 ids
    assert d["config"]["pages_per_gpu_per_pass"] == 8 and "SHARE one GPU" in d["config"]["parallelism"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1      # an N > 1 line carries the CPU leg too (rank 0, behind the timed region)
    assert len(d["ranks"]) == 2
    assert len(d["per_rank_pages_per_s"]["by_rank"]) == 2 and d["per_rank_pages_per_s"]["min"] <= d["per_rank_pages_per_s"]["max"]


def test_bench_five_ranks_share_one_gpu(weights):
    """The same pre-flight at the widest world this pool lets one GPU carry inside the test suite (its process guard ends a run with more than six processes on
    the card - the test process holds the GPU too - so the driver's eight cannot be rehearsed here; six ranks ran from a bare shell, gpurun_out/n6_share_gpu.json):
    five-rank rendezvous, the `flock`ed build, ONE /dev/shm page set with five readers, header / counts / payload gathers at world 5 for every streamed pass,
    `ranks` and the per-rank rates with five entries.  Mirrors the six-thread fan-out of /root/reference/tuatara.cpp:461-475 one
    level up (whole ranks instead of threads).  The line is kept under gpurun_out/ for the record."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(TUATARA_BENCH_SHARE_GPU="1", TUATARA_PRELOAD_TORCH="0", TUATARA_COMM_TIMEOUT="300")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--steps", "2", "--warmup", "1", "--pages", "8", "--buffers", "3", "--deadline", "1200",
                          "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=1300, env=env)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "n5_share_gpu.json"), "w") as f:
            f.write(lines[0] + "\n")
    except OSError:
        pass
    assert d["n_gpus"] == 5 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["gathered_id_rows_last_pass"] == 5 * 8 * 40
    assert len(d["ranks"]) == 5 and sorted(int(r["rank"]) for r in d["ranks"]) == list(range(5))
    assert len(d["per_rank_pages_per_s"]["by_rank"]) == 5
    assert abs(d["value"] - 5 * 8 * 2 / (d["ms_per_step"] * 2 * 1e-3)) / d["value"] < 1e-6      # value = all ranks' pages / the slowest rank's time
