"""Whole-iteration hipGraph capture for small workloads (not in the reference).

A batch that cannot keep the GPU busy is bound by the host: BASELINE config 2 (16 images @512^2) spends 0.19-0.23 ms per eager
step for 0.16 ms of kernels, every launch paying Python, autograd and ctypes.  Nothing on this package's path allocates at the
C-ABI level or synchronises the host (below the rasterizer's scratch limit, include/nvdr_hip.h), so an entire iteration --
forward, backward, optimizer step -- records into ONE hipGraph with stock ``torch.cuda.graph``; this module only packages the
recipe (warm-up on a side stream, capture, replay) so that it is one line in a training loop:

    step = StepGraph(lambda: train_step())      # runs train_step() three times, then records it
    for it in range(iters):
        step()                                   # one graph launch per iteration

Rules of ``torch.cuda.graph`` apply: the recorded call must be static (same shapes, same tensors: update inputs in place with
``copy_``), must not read results back (``.item()``, ``print``) and must use capturable optimizers (``torch.optim.Adam(...,
capturable=True)``).  Measured: config 2 0.226 -> 0.157 ms/step, the config-5 stand-in 730 -> 1120 iterations/s (bench.py).
"""
import torch

__all__ = ["StepGraph"]


class StepGraph:
    def __init__(self, step, warmup=3, device=None):
        """Runs ``step()`` ``warmup`` times on a side stream (allocations, autotuned state and the rasterizer's scratch reach
        their final sizes there), then records one more call.  ``step`` may return tensors: ``self.outputs`` holds what the
        recorded call returned (static tensors, refreshed by every replay)."""
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(1, int(warmup))):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.outputs = step()

    def replay(self):
        self.graph.replay()
        return self.outputs

    __call__ = replay
