"""CPU suite: the N>1 plumbing with world_size 2 over gloo (the same code runs over RCCL in bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tuatara_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_recognise(crops):
    # deterministic stand-in for PARSeq: ids derived from the crop bytes
    s = crops.reshape(len(crops), 32 * 128 * 3).astype(np.int64).sum(1)
    return ((s[:, None] + np.arange(26)[None, :]) % 95).astype(np.int32)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # throughput mode: each rank "processes" its pages, records are all-gathered
        n_pages = 5
        mine = D.pages_of_rank(n_pages, rank, world)
        results = [[{"ids": [(p * 7 + c + k) % 95 for k in range(26)]} for c in range(p + 1)] for p in mine]
        while len(results) < -(-n_pages // world):
            results.append([])                       # ragged: pad with an empty page
        rec = D.pack_records(results, max_crops=8)
        allrec = D.all_gather_records(rec)
        assert allrec.shape == (world, -(-n_pages // world), 8, 26)
        for r in range(world):
            for i, p in enumerate(D.pages_of_rank(n_pages, r, world)):
                for c in range(p + 1):
                    assert allrec[r, i, c].tolist() == [(p * 7 + c + k) % 95 for k in range(26)]
                assert (allrec[r, i, p + 1:] == -1).all()
        # latency mode: crop batch sharded over ranks, ids gathered; ragged (N not divisible) and empty
        for n in (7, 2, 1, 0):
            crops = np.random.default_rng(n).integers(0, 256, (n, 32, 128, 3), dtype=np.uint8)
            ids = D.recognise_sharded(crops if rank == 0 else None, _fake_recognise)
            assert ids.shape == (n, 26)
            assert np.array_equal(ids, _fake_recognise(crops).reshape(n, 26))
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_gather_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert out == [(0, "ok"), (1, "ok")], out


def test_shard_helpers():
    assert D.pages_of_rank(10, 1, 4) == [1, 5, 9]
    assert [D.crop_shard(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [D.crop_shard(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert D.crop_shard(0, 0, 2) == (0, 0)
    rec = D.pack_records([[{"ids": list(range(26))}], []], max_crops=3)
    assert rec.shape == (2, 3, 26) and (rec[1] == -1).all()
    assert D.unpack_records(rec) == [[list(range(26))], []]
