"""Multi-GPU layout of the hot path: one process per GPU, minibatch items sharded.

Every kernel of the path indexes the minibatch item independently (reference: blockIdx.z,
e.g. csrc/common/rasterize.cu:20, interpolate.cu:20), so items shard across GPUs with no
collective on the data path.  What does cross GPUs (new functionality):

  broadcast_shared        geometry/assets every item shares (tri, shared attr/uv, texture,
                          topology hash) from rank 0 -- once, outside the step;
  allreduce_shared_grads  gradients of those shared inputs (sum over ranks) -- the only
                          per-step exchange of a training step;
  gather_items            optional all-gather of per-item images [N/G,...] -> [N,...] when one
                          rank needs the whole batch (serving / visualisation).

Backend "nccl" is RCCL over xGMI on ROCm; the same code runs on gloo for the CPU tests.
"""
import torch
import torch.distributed as dist


_FORCE = False


def force_collectives(enable=True):
    """Testing aid: issue the collectives even in a process group of one (lets a 1-GPU box execute the RCCL calls)."""
    global _FORCE
    _FORCE = bool(enable)


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _single():
    """True when there is nobody to talk to (and the collectives are not forced for testing)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size() == 1 and not _FORCE


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_range(n_items, world_size=None, r=None):
    """Contiguous split of n_items over the ranks -> (start, count) of rank r."""
    world_size = world() if world_size is None else world_size
    r = rank() if r is None else r
    base, rem = divmod(n_items, world_size)
    start = r * base + min(r, rem)
    return start, base + (1 if r < rem else 0)


def shard_items(t, dim=0):
    """This rank's slice of a per-item tensor."""
    s, c = shard_range(t.shape[dim])
    return t.narrow(dim, s, c)


def broadcast_shared(tensors, src=0):
    """In-place broadcast of shared (non per-item) tensors from ``src``."""
    if _single():
        return tensors
    for t in tensors:
        dist.broadcast(t, src=src)
    return tensors


def allreduce_shared_grads(params):
    """Sum the gradients of shared inputs over ranks (one flat bucket, one collective)."""
    if _single():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    if len(grads) == 1:
        dist.all_reduce(grads[0], op=dist.ReduceOp.SUM)
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def gather_items(local, n_items=None):
    """All-gather per-item tensors along dim 0 (ragged splits are padded to the largest shard).

    ``n_items`` = size of the whole batch when it was split with ``shard_range``; when it is not given the
    ranks first exchange their local counts, so that every rank sizes the collective identically whatever the split."""
    w = world()
    if _single():
        return local
    if n_items is not None:
        counts = [shard_range(n_items, w, r)[1] for r in range(w)]
    else:
        mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        every = torch.empty(w, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(every, mine)
        counts = [int(c) for c in every.tolist()]
    mx = max(counts)
    if local.shape[0] < mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((w * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    if all(c == mx for c in counts):
        return out
    return torch.cat([out[r * mx:r * mx + counts[r]] for r in range(w)], 0)


def gather_items_async(local, out=None):
    """Non-blocking all-gather of EQUAL-sized per-item shards along dim 0 -> (work, gathered).

    The collective is issued on the backend's own stream (RCCL: ordered after the kernels that produced ``local``
    on the current stream), so kernels launched afterwards overlap it; ``work.wait()`` makes the current stream
    wait for the result.  ``out`` = receive buffer of a previous call to reuse.  World size 1: (no-op work, local)."""
    w = world()
    if _single():
        return _Done(), local
    shape = (w * local.shape[0],) + tuple(local.shape[1:])
    if out is None or tuple(out.shape) != shape or out.dtype != local.dtype or out.device != local.device:
        out = torch.empty(shape, dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, local.contiguous(), async_op=True)
    return work, out


class _Done:
    def wait(self):
        return True
