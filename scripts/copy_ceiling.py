#!/usr/bin/env python3
"""What a plain device-to-device copy sustains on this box (read + write bytes / time): the practical ceiling the
HBM-bound kernels (Winograd transforms, 1x1 convs with a residual) are compared with.  Run on an MI355X."""
import torch

for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    y = torch.empty_like(x)
    for fn, name, factor in ((lambda: y.copy_(x), "copy", 2), (lambda: y.add_(1.0), "rmw add", 2), (lambda: x.sum(), "read (sum)", 1),
                             (lambda: y.fill_(1.0), "write (fill)", 1)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{mb:5d} MB {name:12s} {factor * mb / 1024 / (ms / 1e3) / 1e3 * 1.073741824:7.2f} TB/s  ({ms * 1e3:.1f} us)")
