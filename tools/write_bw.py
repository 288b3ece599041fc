"""Measurement tool: what a pure write stream of k_fine's size (512 MiB = rast + rast_db at the headline config)
and a pure read+write stream cost on this GPU -- the practical floors behind the per-kernel roofline fractions."""
import torch
dev = torch.device("cuda", 0)
n = 512 << 20
a = torch.empty(n // 4, dtype=torch.float32, device=dev)
b = torch.empty(n // 4, dtype=torch.float32, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t = timed(lambda: a.zero_())
print("fill 512 MiB: %.1f us = %.2f TB/s written" % (t * 1e3, n / t / 1e9))
t = timed(lambda: b.copy_(a))
print("copy 512 MiB: %.1f us = %.2f TB/s read+written" % (t * 1e3, 2 * n / t / 1e9))
t = timed(lambda: a.sum())
print("reduce 512 MiB: %.1f us = %.2f TB/s read" % (t * 1e3, n / t / 1e9))
