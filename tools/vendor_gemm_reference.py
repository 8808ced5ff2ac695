#!/usr/bin/env python3
"""Known-good reference for the NT GEMM shapes (cdna_hip_programming.md, methodology rule 10: a ceiling claim needs a
reference measured on the same hardware): torch.nn.functional.linear in bf16 = the vendor library (hipBLASLt / rocBLAS) on
the same random operands, next to this repository's kernel through the C ABI.  Measurement only: the product never calls it.
    python tools/vendor_gemm_reference.py [--rounds 7]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tests.util import BF16, check, lib, stream, _ptr  # noqa: E402

SHAPES = [("enc qkv", 72000, 1152, 384), ("enc proj", 72000, 384, 384), ("enc fc1 (no epilogue)", 72000, 1536, 384), ("enc fc2", 72000, 384, 1536),
          ("enc qkv dgrad", 72000, 384, 1152), ("dec qkv", 290000, 576, 192), ("dec fc1 (no epilogue)", 290000, 768, 192), ("dec fc2", 290000, 192, 768),
          ("vit-b qkv", 25088, 2304, 768), ("vit-b fc2", 25088, 768, 3072), ("long K", 32768, 1152, 4096)]
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for name, M, N, K in SHAPES:
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    bias = torch.rand(N, device=dev, generator=g)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

    def ours():
        check(lib.mae_linear_fwd(_ptr(A), _ptr(W), _ptr(bias), M, N, K, BF16, 0, BF16, _ptr(out), None, None, stream(dev)))

    def vendor():
        return torch.nn.functional.linear(A, W, bias.to(torch.bfloat16))

    def vendor_nobias():
        return torch.nn.functional.linear(A, W)
    t = {}
    for fn in (ours, vendor, vendor_nobias):
        fn(); fn()
    torch.cuda.synchronize()
    for fn in (ours, vendor, vendor_nobias):
        ts = []
        for _ in range(a.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        t[fn.__name__] = sorted(ts)[len(ts) // 2]
    ref = vendor().float()
    err = float((out.float() - ref).norm() / ref.norm())
    fl = 2 * M * N * K
    print(f"{name:24s} M={M} N={N} K={K} | ours {t['ours']:7.1f} us {fl / t['ours'] / 1e6:6.0f} TF/s | vendor+bias {t['vendor']:7.1f} us {fl / t['vendor'] / 1e6:6.0f} TF/s"
          f" | vendor {t['vendor_nobias']:7.1f} us {fl / t['vendor_nobias'] / 1e6:6.0f} TF/s | rel diff {err:.1e}", flush=True)
