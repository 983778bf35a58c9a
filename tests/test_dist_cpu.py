"""CPU suite: the N > 1 schedules with world_size 2 over gloo - the record framing (counts first, then the payload; a page with more
than 128 crops; an empty page; ragged totals) and latency mode with the REAL fp32 oracle recogniser per rank on FUNSD crops - and the
C++ host's framing logic (ttr_gather_layout, no GPU needed) against the same numpy restatement."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from tuatara_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ids_of(rank, page, crop):
    return [(rank * 31 + page * 7 + crop + k) % 95 for k in range(26)]


def _worker(rank, world, port, q, crops_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # throughput mode: variable-length records; rank 0 has a 150-crop page (nothing is cut at 128) and an empty one
        per_page = [[150, 0, 3], [2, 5, 1]][rank]
        results = [[{"ids": _ids_of(rank, p, c)} for c in range(n)] for p, n in enumerate(per_page)]
        counts, ids = D.frame_records(results)
        counts_all, ids_all = D.all_gather_var(counts, ids)
        assert counts_all.tolist() == [[150, 0, 3], [2, 5, 1]]
        cap, total, first = D.gather_layout(counts_all)
        assert cap == 153 and total.tolist() == [153, 8] and len(ids_all) == 161
        for r in range(world):
            for p in range(3):
                for c in range(counts_all[r, p]):
                    assert ids_all[first[r * 3 + p] + c].tolist() == _ids_of(r, p, c)
        # the {status, pages} header: a rank with another page count, then a rank that failed, fail the batch on BOTH ranks; the next batch works
        for kind in ("pages", "failed"):
            try:
                if kind == "pages":
                    D.all_gather_var(counts[: 3 - rank], ids[: int(counts[: 3 - rank].sum())])
                else:
                    D.all_gather_var(counts, ids, failed=(rank == 1))
                raise AssertionError(kind + " went through")
            except RuntimeError as ex:
                assert "multi-GPU batch" in str(ex)
        counts_all2, ids_all2 = D.all_gather_var(counts, ids)
        assert np.array_equal(counts_all2, counts_all) and np.array_equal(ids_all2, ids_all)
        # latency mode: the fp32 oracle PARSeq recognises this rank's shard of real crops; every rank ends with the whole batch's ids
        from oracle import pipeline
        from tuatara_amd import weights as W
        _, parseq = pipeline.load_models(W.synth_craft(0, True), W.synth_parseq(0))
        crops = np.load(crops_path)["crops"]

        def recognise(c):
            return pipeline.parseq_logits(parseq, c).argmax(-1).astype(np.int32)

        for n in (len(crops), 1, 0):
            ids = D.recognise_sharded(crops[:n] if rank == 0 else None, recognise)
            assert ids.shape == (n, 26)
            if rank == 1 and n:
                q.put(("ids", n, ids.tolist()))
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_framing_and_latency_mode_with_the_oracle_recogniser(tmp_path, oracle_models, funsd):
    from oracle import pipeline, post
    craft, parseq = oracle_models
    d = pipeline.image_to_data(craft, parseq, funsd[:420], debug=True)       # the top of the FUNSD page: a few seconds of CPU
    crops = d["crops"][:7]                                                   # 7 crops: ragged shards (4 + 3)
    assert len(crops) == 7
    single = pipeline.parseq_logits(parseq, crops).argmax(-1)
    path = str(tmp_path / "crops.npz")
    np.savez(path, crops=crops)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, path)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(g for g in got if g[0] in (0, 1)) == [(0, "ok"), (1, "ok")], got
    for g in got:
        if g[0] == "ids":
            n, ids = g[1], np.array(g[2])
            assert np.array_equal(ids, single[:n])                              # gathered == single process
            strs = [post.decode_ids(r) for r in ids]
            assert strs == post.decode_logits(pipeline.parseq_logits(parseq, crops[:n]))[0]


def test_shard_helpers():
    assert D.pages_of_rank(10, 1, 4) == [1, 5, 9]
    assert [D.crop_shard(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [D.crop_shard(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert D.crop_shard(0, 0, 2) == (0, 0)
    counts, ids = D.frame_records([[{"ids": list(range(26))}], []])
    assert counts.tolist() == [1, 0] and ids.shape == (1, 26)


def test_cpp_gather_layout_matches_the_restatement():
    """ttr_gather_layout is the framing the C++ host uses for the RCCL gather (GatherLayout, engine.h): host logic, runs without a GPU."""
    from tuatara_amd.build import build_lib
    from tuatara_amd.engine import gather_layout
    build_lib()
    rng = np.random.default_rng(0)
    for world, pages in ((1, 1), (2, 3), (8, 32), (4, 0)):
        counts = rng.integers(0, 200, (world, pages)).astype(np.int32)
        if pages:
            counts[0, 0] = 300                                                 # far beyond the old 128-crop record
        cap, total, first = gather_layout(counts)
        cap2, total2, first2 = D.gather_layout(counts)
        assert cap == cap2 and total.tolist() == total2.tolist() and first.tolist() == first2.tolist()
