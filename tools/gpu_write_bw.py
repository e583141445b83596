#!/usr/bin/env python3
"""Achievable HBM write / copy rates on this GPU with plain torch kernels (fill_, copy_), for scale next to the frame
kernels' rates (the roofline in bench.py prices against the 8 TB/s peak of MI355X_MICROARCH.md)."""
import torch
x = torch.empty(6 * 1024 ** 3, dtype=torch.uint8, device="cuda:0")
y = torch.empty_like(x)
def rate(fn, nbytes, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return nbytes * n / (e0.elapsed_time(e1) * 1e-3) / 1e12
print("fill_  (write only):        %.2f TB/s" % rate(lambda: x.fill_(0), x.numel()))
print("copy_  (read + write):      %.2f TB/s of traffic (%.2f TB/s written)" % (rate(lambda: y.copy_(x), 2 * x.numel()), rate(lambda: y.copy_(x), x.numel())))
xi = x.view(torch.int64)
print("sum    (read only, int64):  %.2f TB/s" % rate(lambda: xi.sum(), x.numel()))
