"""Multi-process path on CPU (gloo, world_size 2): item sharding, shared-geometry broadcast,
shared-gradient all-reduce and the optional image all-gather of nvdiffrast_amd.parallel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nvdiffrast_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # shared geometry comes from rank 0
        tri = torch.arange(12, dtype=torch.int32).reshape(4, 3) if rank == 0 else torch.zeros(4, 3, dtype=torch.int32)
        attr = torch.full((1, 5, 2), 3.0) if rank == 0 else torch.zeros(1, 5, 2)
        parallel.broadcast_shared([tri, attr], src=0)
        assert tri.flatten().tolist() == list(range(12)) and float(attr.sum()) == 30.0

        # items shard contiguously, ragged when n_items % world != 0
        full = torch.arange(n_items * 6, dtype=torch.float32).reshape(n_items, 2, 3)
        mine = parallel.shard_items(full)
        s, c = parallel.shard_range(n_items)
        assert mine.shape[0] == c and torch.equal(mine, full[s:s + c])

        # per-rank "render": a function of the local items; gather gives the whole batch in order
        img = mine * 2.0
        allimg = parallel.gather_items(img, n_items)
        assert torch.equal(allimg, full * 2.0)

        # gradients of shared inputs are summed over ranks (two tensors -> one flat bucket)
        a = torch.nn.Parameter(torch.zeros(3)); b = torch.nn.Parameter(torch.zeros(2, 2))
        a.grad = torch.full((3,), float(rank + 1)); b.grad = torch.full((2, 2), 10.0 * (rank + 1))
        parallel.allreduce_shared_grads([a, b])
        tot = sum(range(1, world + 1))
        assert torch.allclose(a.grad, torch.full((3,), float(tot))) and torch.allclose(b.grad, torch.full((2, 2), 10.0 * tot))
        q.put((rank, "ok"))
    except Exception as e:           # surface the failure in the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [4, 5])
def test_two_rank_gloo(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_helpers_are_noops():
    t = torch.arange(6.0).reshape(3, 2)
    assert parallel.world() == 1 and parallel.rank() == 0
    assert parallel.shard_range(7, 2, 0) == (0, 4) and parallel.shard_range(7, 2, 1) == (4, 3)
    assert parallel.gather_items(t) is t
    assert parallel.broadcast_shared([t])[0] is t
