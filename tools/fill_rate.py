#!/usr/bin/env python3
"""What a write-only kernel can reach on this GPU: torch fills (tensor.zero_(), tensor.fill_()) and a copy over buffers
far larger than the caches, timed with events.  The roofline of the frame kernels is priced against the HBM peak of
MI355X_MICROARCH.md (8 TB/s); this prints the ceiling a pure store stream measures, for DESIGN.md section 4."""
import json
import torch

dev = torch.device("cuda:0")
out = {}
for gb in (1, 8):
    n = gb * (1 << 30)
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    y = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, fn, bytes_ in (("zero_", lambda: x.zero_(), n), ("fill_", lambda: x.fill_(7), n), ("copy_", lambda: y.copy_(x), 2 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 20
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[f"{name}{gb}GiB_GBs"] = round(bytes_ / ms / 1e6, 1)
    del x, y
print(json.dumps(out))
