#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels on the pretrain-step shapes (GPU box).   python tools/attn_bench.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402
from tests.util import BF16, check, lib, stream, _ptr  # noqa: E402

SHAPES = [("enc  T=36  hd=64", 2000, 36, 6, 64), ("dec  T=145 hd=32", 2000, 145, 6, 32), ("2b   T=145 hd=64", 1000, 145, 8, 64),
          ("vitb enc T=50 hd=64", 512, 50, 12, 64), ("vitb dec T=197 hd=32", 512, 197, 16, 32), ("vitl T=257 hd=64", 256, 257, 16, 64)]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for name, B, T, H, hd in SHAPES:
    qkv = (torch.randn(B, T, 3, H, hd, device=dev, generator=g)).to(torch.bfloat16)
    do = (torch.randn(B, T, H * hd, device=dev, generator=g)).to(torch.bfloat16)
    out = torch.empty(B, T, H * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, T, device=dev)
    dqkv = torch.empty_like(qkv)

    def fwd():
        check(lib.mae_attention_fwd(_ptr(qkv), B, T, H, hd, BF16, _ptr(out), _ptr(lse), stream(dev)))

    def bwd():
        check(lib.mae_attention_bwd(_ptr(qkv), _ptr(out), _ptr(do), _ptr(lse), B, T, H, hd, BF16, _ptr(dqkv), stream(dev)))
    res = {}
    for nm, fn in (("fwd", fwd), ("bwd", bwd)):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[nm] = sorted(ts)[2]
    nb_f = qkv.numel() * 2 + out.numel() * 2
    nb_b = qkv.numel() * 4 + out.numel() * 4
    print(f"{name} B={B}: fwd {res['fwd']:7.1f} us ({nb_f / res['fwd'] / 1e3:5.0f} GB/s)  bwd {res['bwd']:7.1f} us ({nb_b / res['bwd'] / 1e3:5.0f} GB/s)", flush=True)
