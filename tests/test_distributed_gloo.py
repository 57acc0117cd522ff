"""N>1 path on CPU: world_size-2 gloo processes exercise the shard / scatter / gather /
key-broadcast plumbing that one-process-per-GPU runs use with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rs_tfhe_amd import distributed as D
        from rs_tfhe_amd.key import CloudKey
        from rs_tfhe_amd.params import SecurityParams

        count, width = 37, 9
        full = torch.arange(count * width, dtype=torch.int32).reshape(count, width) if rank == 0 else None
        shard = D.scatter_batch(full, count, width, src=0)
        lo, hi = D.shard_range(count, rank, world)
        ok = shard.shape == (hi - lo, width) and int(shard[0, 0]) == lo * width
        # "compute" on the shard (stand-in for the sharded bootstrap): no collective involved
        out = shard * 3 + 1
        gathered = D.gather_batch(out, count, width, dst=0)
        if rank == 0:
            ok = ok and torch.equal(gathered, full * 3 + 1)
        else:
            ok = ok and gathered is None
        # key replication
        P = SecurityParams("TOY", 1, 4, 1, 6, 2, 2, 1e-5, 1e-8)
        ck = None
        if rank == 0:
            rng = np.random.default_rng(0)
            ck = CloudKey(P, rng.standard_normal((4, 2, 2, 1024)),
                          rng.integers(0, 2**32, (1024, 2, 4, 5), dtype=np.uint64).astype(np.uint32))
        rep = D.broadcast_cloud_key(ck, P, src=0)
        rng = np.random.default_rng(0)
        ok = ok and np.array_equal(rep.bootstrapping_key, rng.standard_normal((4, 2, 2, 1024)))
        ok = ok and rep.decomposition_offset == 0x80000000 and rep.key_switching_key.shape == (1024, 2, 4, 5)
        # in-place replication of the engine-layout key buffers (bench.py's multi-rank path; RCCL on GPUs)
        class FakeEngine:
            def __init__(self, fill):
                g = torch.Generator().manual_seed(5)
                self.bufs = [torch.randint(0, 255, (n,), dtype=torch.uint8, generator=g) if fill
                             else torch.zeros(n, dtype=torch.uint8) for n in (4096, 1024, 64)]
                self.off = 0x82080000 if fill else 0
                self.adopted = None
                self.synced = 0

            def synchronize(self):  # receivers drain their engine before their key buffers are overwritten
                self.synced += 1

            def cloud_key_device_tensors(self):
                return self.bufs[0], self.bufs[1], self.bufs[2], self.off

            def adopt_cloud_key(self, off):
                self.adopted = off

        fe = FakeEngine(rank == 0)
        D.broadcast_engine_key(fe, src=0)
        want = FakeEngine(True)
        ok = ok and all(torch.equal(a, b) for a, b in zip(fe.bufs, want.bufs))
        ok = ok and fe.adopted == (None if rank == 0 else 0x82080000)
        ok = ok and fe.synced == (0 if rank == 0 else 1)
        # barrier + max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t) == world
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_shard_scatter_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
