"""Batch sharding over the GPUs of one node: one process per GPU.

The path shards embarrassingly (independent ciphertexts, shared read-only
cloud key: src/parallel/rayon_impl.rs:40-47 is an order-preserving par_map),
so there is NO data-path collective.  torch.distributed ("nccl" = RCCL over
xGMI on GPUs, "gloo" in the CPU tests) is used only for the trivial
scatter of inputs / gather of outputs when a batch lives on one rank, and for
the key broadcast.
"""
from __future__ import annotations

import numpy as np


def shard_range(count: int, rank: int, world: int) -> tuple[int, int]:
    """Static contiguous split [lo, hi) of `count` items for `rank`; order-preserving."""
    base, rem = divmod(count, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_counts(count: int, world: int) -> list[int]:
    return [shard_range(count, r, world)[1] - shard_range(count, r, world)[0] for r in range(world)]


def scatter_batch(full, count: int, width: int, src: int = 0, device=None):
    """Rank `src` holds full [count][width] int32 tensor; every rank gets its shard."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    counts = shard_counts(count, world)
    out = torch.empty((counts[rank], width), dtype=torch.int32, device=device)
    if world == 1:
        out.copy_(full)
        return out
    # point-to-point (xGMI is point-to-point: one send per peer, all links in parallel)
    reqs = []
    if rank == src:
        for r in range(world):
            lo, hi = shard_range(count, r, world)
            if r == src:
                out.copy_(full[lo:hi])
            elif hi > lo:
                reqs.append(dist.isend(full[lo:hi].contiguous(), dst=r))
    elif counts[rank]:
        reqs.append(dist.irecv(out, src=src))
    for q in reqs:
        q.wait()
    return out


def gather_batch(shard, count: int, width: int, dst: int = 0, device=None):
    """Inverse of scatter_batch: rank `dst` returns the full tensor (others None)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1:
        return shard
    reqs = []
    full = None
    if rank == dst:
        full = torch.empty((count, width), dtype=torch.int32, device=device)
        for r in range(world):
            lo, hi = shard_range(count, r, world)
            if r == dst:
                full[lo:hi].copy_(shard)
            elif hi > lo:
                reqs.append((dist.irecv(full[lo:hi], src=r)))
    elif shard.shape[0]:
        reqs.append(dist.isend(shard.contiguous(), dst=dst))
    for q in reqs:
        q.wait()
    return full


def broadcast_cloud_key(cloud_key_or_none, params, src: int = 0):
    """Replicate the CloudKey fields from rank `src` (host tensors; each rank
    then uploads to its own GPU)."""
    import torch
    import torch.distributed as dist

    from .key import CloudKey
    from .params import N

    rank = dist.get_rank()
    # RCCL moves device memory only: stage through this rank's GPU under the "nccl" backend
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    shapes = {
        "bootstrapping_key": ((params.n, 2 * params.l, 2, N), torch.float64),
        "key_switching_key": ((N, params.iks_t, params.base, params.n + 1), torch.int32),
        "blind_rotate_testvec": ((2, N), torch.int32),
    }
    fields = {}
    for name, (shape, dt) in shapes.items():
        if rank == src:
            arr = np.ascontiguousarray(getattr(cloud_key_or_none, name))
            t = torch.from_numpy(arr.view(np.int32) if dt == torch.int32 else arr).reshape(shape).clone()
        else:
            t = torch.empty(shape, dtype=dt)
        t = t.to(dev)
        dist.broadcast(t, src=src)
        t = t.cpu()
        fields[name] = t.numpy().view(np.uint32) if dt == torch.int32 else t.numpy()
    off = torch.tensor([int(cloud_key_or_none.decomposition_offset) if rank == src else 0], dtype=torch.int64).to(dev)
    dist.broadcast(off, src=src)
    off = off.cpu()
    return CloudKey(params, fields["bootstrapping_key"], fields["key_switching_key"], int(off.item()),
                    fields["blind_rotate_testvec"])


def broadcast_engine_key(engine, src: int = 0, in_place: bool = None) -> None:
    """Replicate the cloud key held by rank `src`'s engine into every other rank's engine in the engine layouts
    (nothing is converted twice): RCCL broadcast over xGMI under the "nccl" backend, device to device.

    The transport is chosen ONCE for all ranks, before the first collective (a per-rank fallback would leave the
    ranks issuing different numbers of collectives): `in_place` (default: env TFHE_BCAST_INPLACE == "1") broadcasts
    on the context's own buffers; otherwise -- the default -- each buffer goes through a tensor from torch's
    allocator (one extra device-side copy of 172 MB, once per job), which every RCCL transport accepts.  Receivers
    drain their engine first and hold no valid key until `adopt_cloud_key`."""
    import os

    import torch
    import torch.distributed as dist

    if in_place is None:
        in_place = os.environ.get("TFHE_BCAST_INPLACE") == "1"
    rank = dist.get_rank()
    if rank != src:
        engine.synchronize()  # nothing queued on this engine may still read the buffers about to be overwritten
    bsk, ksk, tv, off = engine.cloud_key_device_tensors()
    offt = torch.tensor([off], dtype=torch.int64, device=bsk.device)
    if dist.get_backend() == "nccl":
        for t in (bsk, ksk, tv, offt):
            if in_place:
                dist.broadcast(t, src=src)  # on the context's own buffer
            else:
                st = t.clone() if rank == src else torch.empty_like(t)
                dist.broadcast(st, src=src)
                if rank != src:
                    t.copy_(st)
                del st
    else:  # gloo moves host memory: stage (plumbing tests on boxes with fewer GPUs than ranks)
        for t in (bsk, ksk, tv, offt):
            h = t.cpu()
            dist.broadcast(h, src=src)
            if rank != src:
                t.copy_(h)
    if bsk.is_cuda:
        torch.cuda.synchronize(bsk.device)
    if rank != src:
        engine.adopt_cloud_key(int(offt.item()))


def sharded_batch_gate(engine, gate: int, a_shard, b_shard, out_shard, stream=None) -> None:
    """Each rank bootstraps its own shard; no exchange inside the computation."""
    engine.batch_gate_dev(gate, a_shard, b_shard, out_shard, stream)
