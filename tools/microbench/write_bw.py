"""Streaming-write bandwidth of the box (what bounds the full-output mode from below): fill / copy of buffers the size
of the sweep's output lists, HIP-event timed.  Run on the GPU box."""
import torch

dev = torch.device("cuda", 0)
n = 3_374_124_288 // 8
x = torch.empty(n, dtype=torch.float64, device=dev)
y = torch.empty(n, dtype=torch.float64, device=dev)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t = timed(lambda: x.fill_(1.0))
print(f"fill  {n * 8 / 1e9:.2f} GB: {t:.3f} ms  -> {n * 8 / t / 1e9:.2f} TB/s written")
t = timed(lambda: x.zero_())
print(f"zero  {n * 8 / 1e9:.2f} GB: {t:.3f} ms  -> {n * 8 / t / 1e9:.2f} TB/s written")
t = timed(lambda: y.copy_(x))
print(f"copy  {n * 8 / 1e9:.2f} GB: {t:.3f} ms  -> {2 * n * 8 / t / 1e9:.2f} TB/s read+written")
