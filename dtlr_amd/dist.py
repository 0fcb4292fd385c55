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


def init_from_env(backend: str | None = None, force_group: bool = False) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's env; initialises the process group if world > 1 (or, force_group, also for a
    single rank: the world-size-1 RCCL smoke test runs the very collectives the 8-GPU job runs)."""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force_group) and not dist.is_initialized():
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


def all_gather_records(labels: torch.Tensor, lengths: torch.Tensor, n_total: int, force_collective: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Gather per-rank decode records ([b_r, nq] int32, [b_r] int32) from contiguous shards into the
    global order.  Ragged shards are padded to the largest shard so ONE all_gather suffices.  A single rank returns its own
    records untouched unless force_collective (tests: the collective path on a one-rank group)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
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


# ---- host-side placement: one Python process per GPU, each issuing ~200 launches per 10 ms step --------------------------------
def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]."""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def split_cpus(cpus: Sequence[int], slot: int, n_slots: int) -> List[int]:
    """The slot-th of n_slots contiguous, near-equal parts of `cpus` (never empty while len(cpus) >= 1)."""
    cpus = sorted(cpus)
    if n_slots <= 1 or len(cpus) < n_slots:
        return list(cpus)
    lo, hi = shard_bounds(len(cpus), slot, n_slots)
    return cpus[lo:hi]


def gpu_numa_node(local: int) -> int:
    """NUMA node of GPU `local` from sysfs (-1 when the platform does not say)."""
    try:
        p = torch.cuda.get_device_properties(local)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            return int(f.read().strip())
    except Exception:
        return -1


def pin_to_local_cpus(local: int, world: int, max_threads: int = 4):
    """Bind this rank to CPUs next to its GPU: the cpulist of the GPU's NUMA node, divided among the ranks whose GPUs sit on the same
    node (fallback: an equal share of the process's current affinity).  Keeps 8 launch loops from migrating across sockets and from
    sharing cores; torch's intra-op pool is capped (the step has no host-side tensor math).  Returns a description for the bench line."""
    info = {"numa_node": None, "cpus": None}
    try:
        # `world` counts the ranks of the whole job; CPUs are shared among the ranks of THIS node only (a 16-rank job on two nodes
        # must not give each rank 1/16 of a node, nor ask for the NUMA node of a GPU the node does not have)
        n_local = int(os.environ.get("LOCAL_WORLD_SIZE", 0)) or world
        if torch.cuda.is_available():
            n_local = min(n_local, max(torch.cuda.device_count(), 1))
        world = max(1, min(world, n_local))
        local = local % world
        allowed = sorted(os.sched_getaffinity(0))
        node = gpu_numa_node(local)
        cpus, slot, n_slots = allowed, local, world
        if node >= 0:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                near = [c for c in parse_cpulist(f.read()) if c in set(allowed)]
            if near:
                peers = sorted(r for r in range(world) if gpu_numa_node(r) == node)
                cpus, slot, n_slots = near, peers.index(local) if local in peers else 0, max(len(peers), 1)
                info["numa_node"] = node
        mine = split_cpus(cpus, slot, n_slots)
        if mine:
            os.sched_setaffinity(0, mine)
            info["cpus"] = f"{mine[0]}-{mine[-1]} ({len(mine)})"
        if world > 1:
            torch.set_num_threads(max(1, min(max_threads, len(mine) or 1)))
    except Exception as e:                      # placement is an optimisation: never fail the job over it
        info["error"] = repr(e)
    return info


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device, force_collective: bool = False) -> float:
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
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
