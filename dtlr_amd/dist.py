"""Data-parallel inference over the GPUs of one node.

Lines are independent (no cross-sample op anywhere in DINO.forward), so a global batch shards into
contiguous per-rank slices with replicated weights, one process per GPU, and NO data-path
collective.  The only exchange is ONE all-gather of the decoded fixed-width records
(labels[nq] int32 + length int32 ~= 3.6 KB/line) -- RCCL over xGMI on the GPU box
(backend "nccl"), gloo in the CPU tests.  The reference has no inference DP (its DDP is
training-only, finetuning.py:211-215); correctness check = gathered results equal the
single-process results on the same lines, in order.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's env; initialises the process group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank; the first (n % world) ranks take one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(items: Sequence, rank: int, world: int):
    lo, hi = shard_bounds(len(items), rank, world)
    return items[lo:hi]


def all_gather_records(labels: torch.Tensor, lengths: torch.Tensor, n_total: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Gather per-rank decode records ([b_r, nq] int32, [b_r] int32) from contiguous shards into the
    global order.  Ragged shards are padded to the largest shard so ONE all_gather suffices."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return labels, lengths
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-n_total // world)
    nq = labels.shape[1]
    rec = torch.full((per, nq + 1), -1, dtype=torch.int32, device=labels.device)
    b = labels.shape[0]
    rec[:b, :nq] = labels
    rec[:b, nq] = lengths
    if dist.get_backend() == "nccl":                      # RCCL over xGMI: one flat all-gather of device buffers
        out = torch.empty((world, per, nq + 1), dtype=torch.int32, device=labels.device)
        dist.all_gather_into_tensor(out.view(-1), rec.view(-1))
    else:                                                # gloo (CPU tests; two ranks sharing one GPU in the single-GPU DP test)
        host = rec.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        out = torch.stack(parts, 0).to(labels.device)
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        rows.append(out[r, : hi - lo])
    full = torch.cat(rows, 0)
    return full[:, :nq].contiguous(), full[:, nq].contiguous()


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def finalize() -> None:
    """Last barrier + process-group teardown (all ranks call it once, at the very end)."""
    if dist.is_available() and dist.is_initialized():
        try:
            if dist.get_world_size() > 1:
                dist.barrier()
            dist.destroy_process_group()
        except Exception:       # a peer that already left must not turn a finished run into a failure
            pass
