"""Multi-GPU layout of the hot path: one process per GPU, minibatch items sharded.

Every kernel of the path indexes the minibatch item independently (reference: blockIdx.z,
e.g. csrc/common/rasterize.cu:20, interpolate.cu:20), so items shard across GPUs with no
collective on the data path.  What does cross GPUs (new functionality):

  broadcast_shared        geometry/assets every item shares (tri, shared attr/uv, texture,
                          topology hash) from rank 0 -- once, outside the step;
  allreduce_shared_grads  gradients of those shared inputs (sum over ranks) -- the only
                          per-step exchange of a training step;
  gather_items            optional all-gather of per-item images [N/G,...] -> [N,...] when one
                          rank needs the whole batch (serving / visualisation);
  ImageGather             the same every step, in a form the links can carry: the images are packed on the
                          producing rank (f32 as they are, f16, or rgb8 = unorm8: csrc/image_pack.hip), the packed
                          bytes are all-gathered on RCCL's stream, and the result of step s is collected at the end
                          of step s + 1 -- the link time hides behind the next step's kernels.

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


# ---- images in the form the links can carry ----------------------------------------------------------------------------------
# xGMI is point to point: with every rank receiving every other rank's images, each of a rank's links carries one peer's payload,
# whatever the number of GPUs.  At the headline batch (64 items of 512^2 RGBA per GPU) that is 256 MiB per link and step as f32:
# 1.75 ms against 0.33 ms of rendering.  A consumer of rendered images (a logger, a discriminator, a display) needs 8 bits a
# channel: 64 MiB, 0.44 ms -- and one step later is early enough, so the transfer runs behind the next step's kernels.
FORMATS = {"f32": 0, "f16": 1, "rgba8": 2, "rgb8": 2}      # -> include/nvdr_hip.h NVDR_IMAGE_*; rgb8 ships the first three channels only
_PAYLOAD_DTYPE = {"f32": torch.float32, "f16": torch.float16, "rgba8": torch.uint8, "rgb8": torch.uint8}


def payload_channels(fmt, channels):
    return min(channels, 3) if fmt == "rgb8" else channels


def payload_bytes_per_pixel(fmt, channels):
    return payload_channels(fmt, channels) * {"f32": 4, "f16": 2, "rgba8": 1, "rgb8": 1}[fmt]


def pack_images(images, fmt, out=None):
    """f32 images [..., C] -> the payload that travels: f32: `images` itself; f16: halves; rgba8: round(clamp(x, 0, 1) * 255) as
    uint8, every channel; rgb8: the same for the first three channels only ([..., 3]: an RGB image out of RGBA / four
    attributes).  GPU tensors: one streaming kernel of the library (nvdr_image_pack).  CPU tensors -- the gloo tests of this
    module's plumbing, bench.py --dry-run-cpu -- the same arithmetic by torch operations."""
    assert fmt in FORMATS
    if fmt == "f32":
        return images
    assert images.dtype == torch.float32
    src = images.contiguous()
    C = src.shape[-1]
    CO = payload_channels(fmt, C)
    shape = tuple(src.shape[:-1]) + (CO,)
    if out is None or tuple(out.shape) != shape or out.dtype != _PAYLOAD_DTYPE[fmt] or out.device != src.device:
        out = torch.empty(shape, dtype=_PAYLOAD_DTYPE[fmt], device=src.device)
    if src.is_cuda:
        from . import _capi
        with torch.cuda.device(src.device):
            _capi.check(_capi.load().nvdr_image_pack(src.data_ptr(), out.data_ptr(), src.numel() // C, C, CO, FORMATS[fmt],
                                                     torch.cuda.current_stream(src.device).cuda_stream), "image_pack")
    elif fmt == "f16":
        out.copy_(src)
    else:
        out.copy_((src[..., :CO].clamp(0.0, 1.0) * 255.0).round())
    return out


def unpack_images(payload, fmt, out=None):
    """The inverse of pack_images for a receiver that wants f32 again (rgba8 / rgb8: q / 255; rgb8 stays three channels)."""
    assert fmt in FORMATS
    if fmt == "f32":
        return payload
    src = payload.contiguous()
    if out is None or out.shape != src.shape or out.dtype != torch.float32 or out.device != src.device:
        out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    if src.is_cuda:
        from . import _capi
        with torch.cuda.device(src.device):
            _capi.check(_capi.load().nvdr_image_unpack(src.data_ptr(), out.data_ptr(), src.numel(), FORMATS[fmt],
                                                       torch.cuda.current_stream(src.device).cuda_stream), "image_unpack")
    elif fmt == "f16":
        out.copy_(src)
    else:
        out.copy_(src.to(torch.float32) / 255.0)
    return out


class GatherHandle:
    """One image all-gather in flight: `wait()` makes the current stream wait for it and returns the whole batch's payload
    [n_items, ...] in item order (ragged splits: the padding of the smaller shards is dropped); `images()` returns it as f32."""
    __slots__ = ("work", "recv", "counts", "fmt", "_keep")

    def __init__(self, work, recv, counts, fmt, keep):
        self.work, self.recv, self.counts, self.fmt, self._keep = work, recv, counts, fmt, keep

    def wait(self):
        self.work.wait()
        self._keep = None                               # (the send buffer may go now)
        mx = max(self.counts)
        if all(c == mx for c in self.counts):
            return self.recv
        return torch.cat([self.recv[r * mx:r * mx + c] for r, c in enumerate(self.counts)], 0)

    def images(self, out=None):
        return unpack_images(self.wait(), self.fmt, out)


def start_image_gather(local, fmt="f32", n_items=None, recv=None, send=None):
    """Pack this rank's images and start their all-gather on the backend's own stream (ordered after the kernels that wrote
    `local`); kernels launched afterwards overlap it.  `n_items`: size of the whole batch when it was split with shard_range
    (ragged splits are padded to the largest shard); None: equal shards.  `recv`: a receive buffer of an earlier call to reuse
    (the caller must be done with what it last held); `send`: likewise a packed send buffer of a collective that has completed
    (packed formats only) -- with both, a steady loop allocates nothing.  -> GatherHandle."""
    w = world()
    payload = pack_images(local, fmt, out=send)
    if _single():
        return GatherHandle(_Done(), payload, [payload.shape[0]], fmt, None)
    counts = [local.shape[0]] * w if n_items is None else [shard_range(n_items, w, r)[1] for r in range(w)]
    mx = max(counts)
    if payload.shape[0] < mx:
        pad = torch.zeros((mx - payload.shape[0],) + tuple(payload.shape[1:]), dtype=payload.dtype, device=payload.device)
        payload = torch.cat([payload, pad], 0)
    shape = (w * mx,) + tuple(payload.shape[1:])
    if recv is None or tuple(recv.shape) != shape or recv.dtype != payload.dtype or recv.device != payload.device:
        recv = torch.empty(shape, dtype=payload.dtype, device=payload.device)
    payload = payload.contiguous()
    work = dist.all_gather_into_tensor(recv, payload, async_op=True)
    h = GatherHandle(work, recv, counts, fmt, payload)
    return h


class ImageGather:
    """The per-step exchange of output images, one step deep.

        g = ImageGather("rgb8", n_items=total)        # once
        every step:   images = render(...)            # this rank's items, f32 [n, H, W, C]
                      g.submit(images)                # pack + start the all-gather; returns at once
                      ... backward, optimizer ...
                      batch = g.collect()             # the WHOLE batch of the PREVIOUS step (None in the first), waited for here

    `pipelined=False`: collect() returns the batch of the step just submitted (the link time is then part of the step).
    Three receive buffers rotate: what collect() returned stays valid through the whole next step (until the submit() after the
    next collect())."""

    def __init__(self, fmt="rgb8", n_items=None, pipelined=True):
        assert fmt in FORMATS
        self.fmt, self.n_items, self.pipelined = fmt, n_items, bool(pipelined)
        self._recv = [None, None, None]
        self._send = [None, None, None]
        self._step = 0
        self._flying = None          # the handle submit() started and nobody has collected yet
        self._previous = None        # pipelined: the handle of the step before

    def submit(self, images):
        assert self._flying is None, "ImageGather: collect() must follow every submit()"
        k = self._step % 3
        # (buffer k's previous collective -- three submits ago -- has been waited for by the collect() after the next submit)
        self._flying = start_image_gather(images, self.fmt, self.n_items, self._recv[k], self._send[k])
        self._recv[k] = self._flying.recv
        if self.fmt != "f32" and self._flying._keep is not None:
            self._send[k] = self._flying._keep
        self._step += 1

    def collect(self):
        """-> payload of the whole batch ([n_items, ...] in the gather's format) or None (first pipelined step)."""
        h, self._flying = self._flying, None
        if not self.pipelined:
            return None if h is None else h.wait()
        done, self._previous = self._previous, h
        return None if done is None else done.wait()

    def drain(self):
        """The batches still in flight (end of the loop), oldest first."""
        out = []
        for h in (self._previous, self._flying):
            if h is not None:
                out.append(h.wait())
        self._previous = self._flying = None
        return out
