"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm,
"gloo" in the CPU tests).  The OCR path has no data-path collective: pages are independent and,
inside a page, crops are independent.  The only exchange is gathering the decoded
token ids — fixed-size records, a few KB, latency-bound on xGMI — so one all_gather per step.

  throughput mode : page p -> rank p % G; every rank runs the whole pipeline on its pages.
  latency mode    : rank 0 detects and packs crops, broadcasts them, rank r recognises the
                    contiguous shard r of the crop batch, ids are all-gathered.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

MAX_CROPS = 128      # record capacity per page
L = 26               # token ids per crop


def pages_of_rank(n_pages: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_pages, world))


def crop_shard(n_crops: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ceil(N/G) crops; trailing ranks may be empty."""
    per = -(-n_crops // world) if n_crops else 0
    lo = min(n_crops, rank * per)
    return lo, min(n_crops, lo + per)


def pack_records(results: Sequence[Sequence[dict]], max_crops: int = MAX_CROPS) -> np.ndarray:
    """results[page][crop]["ids"] -> int32 [pages, max_crops, 26], -1 padded."""
    rec = np.full((len(results), max_crops, L), -1, np.int32)
    for i, r in enumerate(results):
        ids = getattr(r, "ids", None)
        if isinstance(ids, np.ndarray):            # engine.PageResult: the id array the C ABI filled
            k = min(len(ids), max_crops)
            rec[i, :k] = ids[:k]
            continue
        for j, item in enumerate(r[:max_crops]):
            rec[i, j] = item["ids"]
    return rec


def unpack_records(rec: np.ndarray) -> List[List[List[int]]]:
    out = []
    for page in rec:
        out.append([row.tolist() for row in page if row[0] != -1 or (row != -1).any()])
    return out


def all_gather_records(rec: np.ndarray, device: str = "cpu") -> np.ndarray:
    """[pages, C, 26] on every rank -> [world, pages, C, 26] on every rank (one collective)."""
    import torch
    import torch.distributed as dist

    mine = torch.from_numpy(np.ascontiguousarray(rec)).to(device)
    world = dist.get_world_size()
    out = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=device)
    dist.all_gather_into_tensor(out, mine)   # concatenation along dim 0 (the form gloo and RCCL both accept)
    return out.reshape((world,) + tuple(mine.shape)).cpu().numpy()


def recognise_sharded(crops: np.ndarray | None, recognise: Callable[[np.ndarray], np.ndarray], device: str = "cpu") -> np.ndarray:
    """Latency mode.  rank 0 passes the packed crop batch u8 [N,32,128,3] (others pass None);
    every rank returns the ids int32 [N,26] of the whole batch."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    n = torch.tensor([len(crops) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, 0)
    n = int(n.item())
    if n == 0:
        return np.zeros((0, L), np.int32)
    buf = torch.from_numpy(np.ascontiguousarray(crops)).to(device) if rank == 0 else torch.empty((n, 32, 128, 3), dtype=torch.uint8, device=device)
    if n:
        dist.broadcast(buf, 0)
    per = -(-n // world) if n else 0
    lo, hi = crop_shard(n, rank, world)
    ids = np.full((per, L), -1, np.int32)
    if hi > lo:
        ids[: hi - lo] = recognise(buf[lo:hi].cpu().numpy())
    gathered = all_gather_records(ids[None], device)[:, 0]          # [world, per, 26]
    return gathered.reshape(world * per, L)[:n]
