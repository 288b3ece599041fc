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

        # ---- the per-step image exchange: format x pipelining, on this (possibly ragged) split, three steps --------------------
        torch.manual_seed(7)
        frames = [torch.rand(n_items, 4, 6, 4) * 1.4 - 0.2 for _ in range(3)]          # values below 0 and above 1 as well
        for fmt in ("f32", "f16", "rgba8", "rgb8"):
            want = []
            for f in frames:
                if fmt == "f32":
                    want.append(f)
                elif fmt == "f16":
                    want.append(f.half())
                else:
                    want.append((f[..., :3 if fmt == "rgb8" else 4].clamp(0, 1) * 255).round().to(torch.uint8))
            for pipelined in (True, False):
                g = parallel.ImageGather(fmt, n_items=n_items, pipelined=pipelined)
                got = []
                for f in frames:
                    g.submit(f[s:s + c])
                    batch = g.collect()
                    got.append(None if batch is None else batch.clone())      # (the receive buffers rotate: see below)
                got += g.drain()
                if pipelined:
                    assert got[0] is None and len(got) == 4
                    got = got[1:]
                else:
                    assert len(got) == 3
                for k in range(3):
                    assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), (fmt, pipelined, k)   # bit-exact payload, item order
                back = parallel.unpack_images(got[2], fmt)
                ref = frames[2][..., :back.shape[-1]]
                if fmt == "f32":
                    assert torch.equal(back, ref)
                elif fmt == "f16":
                    assert (back - ref).abs().max() <= 1e-3
                else:
                    assert (back - ref.clamp(0, 1)).abs().max() <= 0.5 / 255 + 1e-7      # half a step of the 8-bit grid
        # three receive buffers rotate: what collect() returned stays valid through the whole next step
        g = parallel.ImageGather("rgba8", n_items=n_items)
        g.submit(frames[0][s:s + c]); assert g.collect() is None
        g.submit(frames[1][s:s + c]); first = g.collect(); keep = first.clone()
        g.submit(frames[2][s:s + c]); second = g.collect()
        assert torch.equal(first, keep) and not torch.equal(second, keep)
        g.submit(frames[0][s:s + c])                                    # (the fourth submit takes the first buffer again)
        last = g.drain()[-1]
        assert n_items % world or first.data_ptr() == last.data_ptr()    # (ragged splits hand out a copy without the padding)
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
    # the image exchange with nobody to talk to: the payload of this rank's own images, pipelined or not
    img = torch.rand(2, 4, 4, 4)
    for fmt in ("f32", "f16", "rgba8", "rgb8"):
        g = parallel.ImageGather(fmt)
        g.submit(img); assert g.collect() is None
        g.submit(img * 0.5)
        assert torch.equal(g.collect(), parallel.pack_images(img, fmt))
        assert torch.equal(g.drain()[0], parallel.pack_images(img * 0.5, fmt))
    assert parallel.payload_bytes_per_pixel("rgb8", 4) == 3 and parallel.payload_bytes_per_pixel("rgba8", 4) == 4
    assert parallel.payload_bytes_per_pixel("f16", 3) == 6 and parallel.payload_bytes_per_pixel("rgb8", 2) == 2
