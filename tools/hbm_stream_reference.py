#!/usr/bin/env python3
"""What PyTorch-ROCm's own streaming kernels reach on this part (GPU box): the yardstick the HBM-bound kernels (LayerNorm, slab
reduce, AdamW) are held against.  Measured on MI355X: torch.add 6.2 TB/s (read + write), fill 6.9 TB/s (write), device copy
5.0 TB/s, sum 4.0 TB/s (read).    python tools/hbm_stream_reference.py"""
import torch
dev = torch.device("cuda:0")
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.rand(n, device=dev); y = torch.empty_like(x)
    for fn, name, factor in ((lambda: y.copy_(x), "copy (r+w)", 2), (lambda: x.sum(), "sum (r)", 1), (lambda: y.fill_(1.0), "fill (w)", 1), (lambda: torch.add(x, 1.0, out=y), "add (r+w)", 2)):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        t = sorted(ts)[3]
        print(f"{mb:5d} MiB {name:10s} {t*1e3:8.1f} us  {factor * n * 4 / t / 1e9:7.2f} TB/s", flush=True)
