"""Multi-GPU launcher glue (one process per GPU).  The exchange itself lives in the C++ host behind the C ABI - RCCL
all-gather of the token ids on the engine's stream, latency mode's crop broadcast (include/tuatara_hip.h, "multi-GPU";
tuatara_amd.engine.Comm) - and needs no torch.  What is left here:

  * how work is dealt out: page p -> rank p % G (throughput mode), contiguous crop shards (latency mode);
  * the same record framing and the same latency-mode schedule over torch.distributed, for the CPU tests (backend "gloo",
    world_size 2: tests/test_dist_cpu.py) - counts first, then the payload, nothing truncated - mirroring
    GatherLayout (engine.h) / Engine::detect_collect and Engine::run_pages_sharded (engine_pages.cpp).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

L = 26               # token ids per crop


def pages_of_rank(n_pages: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_pages, world))


def crop_shard(n_crops: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ceil(N/G) crops; trailing ranks may be empty."""
    per = -(-n_crops // world) if n_crops else 0
    lo = min(n_crops, rank * per)
    return lo, min(n_crops, lo + per)


def frame_records(results: Sequence[Sequence[dict]]) -> Tuple[np.ndarray, np.ndarray]:
    """One rank's batch -> (counts int32 [pages], ids int32 [sum(counts), 26]): any number of crops per page."""
    counts = np.array([len(r) for r in results], np.int32)
    rows = []
    for r in results:
        ids = getattr(r, "ids", None)
        rows.append(np.asarray(ids, np.int32).reshape(-1, L) if isinstance(ids, np.ndarray) else np.array([it["ids"] for it in r], np.int32).reshape(-1, L))
    return counts, (np.concatenate(rows) if rows else np.zeros((0, L), np.int32))


def gather_layout(counts_all: np.ndarray):
    """counts [world, pages] -> (cap, total[world], first[world * pages + 1]): GatherLayout::from_counts restated."""
    counts_all = np.asarray(counts_all, np.int64)
    total = counts_all.sum(1)
    first = np.concatenate([[0], np.cumsum(counts_all.reshape(-1))])
    return int(total.max()) if total.size else 0, total.astype(np.int32), first


def all_gather_var(counts: np.ndarray, ids: np.ndarray, device: str = "cpu", failed: bool = False):
    """Every rank's (counts, ids) -> (counts_all [world, pages], ids_all [sum, 26] in (rank, page, crop) order): a {status, pages} header
    (Engine::detect_collect: a rank that failed before the exchange, or passed another page count, fails the batch on EVERY rank instead of
    leaving the others in a gather of mismatched sizes), then two collectives - the counts, then the payload padded to the largest rank total."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    hdr = torch.tensor([-1 if failed else 0, len(counts)], dtype=torch.int32, device=device)
    hall = torch.empty((world * 2,), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(hall, hdr)
    hall = hall.cpu().numpy().reshape(world, 2)
    if failed:
        raise RuntimeError("multi-GPU batch: this rank failed before the exchange")
    for r in range(world):
        if hall[r, 0] < 0:
            raise RuntimeError(f"multi-GPU batch: rank {r} failed before the exchange; the batch is dropped on every rank")
        if hall[r, 1] != len(counts):
            raise RuntimeError(f"multi-GPU batch: rank {r} passed {hall[r, 1]} pages, this rank {len(counts)}")
    c = torch.from_numpy(np.ascontiguousarray(counts, np.int32)).to(device)
    call = torch.empty((world * c.shape[0],), dtype=c.dtype, device=device)
    dist.all_gather_into_tensor(call, c)
    counts_all = call.cpu().numpy().reshape(world, -1)
    cap, total, _ = gather_layout(counts_all)
    pay = torch.full((max(cap, 1), L), -1, dtype=torch.int32, device=device)
    if len(ids):
        pay[: len(ids)] = torch.from_numpy(np.ascontiguousarray(ids, np.int32)).to(device)
    out = torch.empty((world * pay.shape[0], L), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out, pay)
    out = out.cpu().numpy().reshape(world, -1, L)
    return counts_all, np.concatenate([out[r, : total[r]] for r in range(world)]) if total.sum() else np.zeros((0, L), np.int32)


def recognise_sharded(crops: np.ndarray | None, recognise: Callable[[np.ndarray], np.ndarray], device: str = "cpu") -> np.ndarray:
    """Latency mode over torch.distributed (the CPU twin of ttr_pages_to_data_dev_sharded).  rank 0 passes the packed crop batch
    u8 [N,32,128,3] (others pass None); every rank returns the ids int32 [N,26] of the whole batch."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    n = torch.tensor([len(crops) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, 0)
    n = int(n.item())
    if n == 0:
        return np.zeros((0, L), np.int32)
    buf = torch.from_numpy(np.ascontiguousarray(crops)).to(device) if rank == 0 else torch.empty((n, 32, 128, 3), dtype=torch.uint8, device=device)
    dist.broadcast(buf, 0)
    per = -(-n // world)
    lo, hi = crop_shard(n, rank, world)
    ids = torch.full((per, L), -1, dtype=torch.int32, device=device)
    if hi > lo:
        ids[: hi - lo] = torch.from_numpy(np.ascontiguousarray(recognise(buf[lo:hi].cpu().numpy()), np.int32)).to(device)
    out = torch.empty((world * per, L), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out, ids)
    return out.cpu().numpy()[:n]
