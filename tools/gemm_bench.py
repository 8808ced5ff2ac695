#!/usr/bin/env python3
"""Micro-benchmark of the Linear GEMM kernels on the pretrain-step shapes (GPU box).  Interleaved A/B of the NT kernel
variants in one process (MAE_GEMM_NT=v1|v2), random bf16 data, reports TFLOP/s and GB/s and checks the variants agree
bit for bit.    python tools/gemm_bench.py [--rounds 5]"""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tests.util import BF16, F32, TDT, check, lib, stream, _ptr  # noqa: E402

SHAPES = [  # name, M, N, K, epilogue, out dtype
    ("enc qkv fwd", 72000, 1152, 384, "none", BF16),
    ("enc fc1 fwd (gelu)", 72000, 1536, 384, "gelu", BF16),
    ("enc proj fwd", 72000, 384, 384, "none", BF16),
    ("enc fc2 fwd", 72000, 384, 1536, "none", BF16),
    ("enc fc2 dgrad (dgelu)", 72000, 1536, 384, "dgelu", BF16),
    ("enc fc1 fwd (gelu_grad)", 72000, 1536, 384, "gelu_grad", BF16),
    ("enc fc2 dgrad (mul)", 72000, 1536, 384, "mul", BF16),
    ("tgt fc1 fwd (gelu_act)", 288000, 1536, 384, "gelu_act", BF16),
    ("enc qkv dgrad", 72000, 384, 1152, "none", BF16),
    ("enc proj fwd resid", 72000, 384, 384, "resid", F32),
    ("enc fc2 fwd resid", 72000, 384, 1536, "resid", F32),
    ("dec fc2 fwd resid", 290000, 192, 768, "resid", F32),
    ("dec qkv fwd", 290000, 576, 192, "none", BF16),
    ("dec fc1 fwd (gelu)", 290000, 768, 192, "gelu", BF16),
    ("dec fc2 fwd", 290000, 192, 768, "none", BF16),
    ("dec fc1 fwd (gelu_grad)", 290000, 768, 192, "gelu_grad", BF16),
    ("dec fc2 dgrad (mul)", 290000, 768, 192, "mul", BF16),
    ("dec fc1 dgrad", 290000, 192, 768, "none", BF16),
    ("dec proj fwd", 290000, 192, 192, "none", BF16),
    ("dec qkv dgrad", 290000, 192, 576, "none", BF16),
    ("dec embed", 72000, 192, 384, "none", BF16),
    ("patch embed", 72000, 384, 192, "none", F32),
    ("pred head", 218000, 192, 192, "none", F32),
    # cache-resident probes (operands + output fit the 256 MiB Infinity Cache after the first round)
    ("probe qkv M=32768", 32768, 1152, 384, "none", BF16),
    ("probe fc2 M=32768", 32768, 384, 1536, "none", BF16),
    ("probe fc1 none", 72000, 1536, 384, "none", BF16),
    ("probe K=4096", 32768, 1152, 4096, "none", BF16),
    # ViT-B/16 (batch 512: 25088 encoder rows, 100864 decoder rows at width 512) and ViT-L/14 (65536 rows) shapes
    ("vitb qkv fwd", 25088, 2304, 768, "none", BF16),
    ("vitb fc1 fwd (gelu_grad)", 25088, 3072, 768, "gelu_grad", BF16),
    ("vitb fc2 fwd", 25088, 768, 3072, "none", BF16),
    ("vitb dec fc1 (gelu_grad)", 100864, 2048, 512, "gelu_grad", BF16),
    ("vitb dec fc2 fwd", 100864, 512, 2048, "none", BF16),
    ("vitl fc1 fwd (gelu_grad)", 65536, 4096, 1024, "gelu_grad", BF16),
    ("vitl fc2 fwd", 65536, 1024, 4096, "none", BF16),
]
WGRAD_SHAPES = [  # name, M, N, K   (dW[N,K] = dY[M,N]^T A[M,K], db = colsum dY)
    ("enc qkv wgrad", 72000, 1152, 384), ("enc proj wgrad", 72000, 384, 384), ("enc fc1 wgrad", 72000, 1536, 384),
    ("enc fc2 wgrad", 72000, 384, 1536), ("dec qkv wgrad", 290000, 576, 192), ("dec proj wgrad", 290000, 192, 192),
    ("dec fc1 wgrad", 290000, 768, 192), ("dec fc2 wgrad", 290000, 192, 768), ("patch wgrad", 70000, 384, 192),
    ("dec embed wgrad", 72000, 192, 384), ("pred wgrad", 218000, 192, 192),
    # widths that are not multiples of 192 (ViT-L/14, the 512-wide decoder of configs[3]; ViT-B/16 for comparison)
    ("vitl qkv wgrad", 65536, 3072, 1024), ("vitl proj wgrad", 65536, 1024, 1024), ("vitl fc1 wgrad", 65536, 4096, 1024),
    ("dec512 qkv wgrad", 100864, 1536, 512), ("dec512 fc2 wgrad", 100864, 512, 2048), ("vitb fc1 wgrad", 25088, 3072, 768),
]
MODE = {"none": 0, "gelu": 1, "resid": 2, "dgelu": 3, "gelu_grad": 4, "mul": 5, "gelu_act": 6}


def wgrad_main(args, dev):
    """Interleaved A/B of the wgrad kernel variants (MAE_WGRAD, read per call) with a bitwise comparison against the first."""
    variants = args.wgrad_variants.split(",")
    g = torch.Generator(device=dev).manual_seed(1)
    for name, M, N, K in WGRAD_SHAPES:
        if args.only and not any(o in name for o in args.only.split(',')):
            continue
        dY = (torch.rand(M, N, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        outs = {v: (torch.empty(N, K, device=dev), torch.empty(N, device=dev)) for v in variants}
        scratch = torch.empty(max(1, lib.mae_linear_wgrad_scratch_bytes(M, N, K)), dtype=torch.uint8, device=dev)

        def run(v):
            os.environ["MAE_WGRAD"] = v
            dW, db = outs[v]
            check(lib.mae_linear_wgrad(_ptr(dY), _ptr(A), M, N, K, BF16, _ptr(dW), _ptr(db), _ptr(scratch), stream(dev)))
        for v in variants:
            run(v)
        torch.cuda.synchronize()
        times = {v: [] for v in variants}
        for _ in range(args.rounds):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(v); e1.record(); torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3)
        line = f"{name:18s} M={M} N={N} K={K}"
        for v in variants:
            t = sorted(times[v])[len(times[v]) // 2]
            same = "" if v == variants[0] else (" ==" if torch.equal(outs[v][0], outs[variants[0]][0]) and torch.equal(outs[v][1], outs[variants[0]][1]) else " DIFFERS")
            line += f" | {v}: {t:7.1f} us {2 * M * N * K / t / 1e6:5.0f} TF/s{same}"
        print(line, flush=True)
    os.environ.pop("MAE_WGRAD", None)


PAIRS = [  # name, M, (N0, K0), (N1, K1): the engine's pairings of a block's weight gradients
    ("enc fc2 + fc1", 72000, (384, 1536), (1536, 384)), ("enc proj + qkv", 72000, (384, 384), (1152, 384)),
    ("dec fc2 + fc1", 290000, (192, 768), (768, 192)), ("dec proj + qkv", 290000, (192, 192), (576, 192)),
    ("vitb fc2 + fc1", 25088, (768, 3072), (3072, 768)), ("vitb proj + qkv", 25088, (768, 768), (2304, 768)),
]


def wgrad_pair_main(args, dev):
    """One paired launch (mae_linear_wgrad_pair) against the two separate launches it replaces (MAE_WGRAD_PAIR=0)."""
    g = torch.Generator(device=dev).manual_seed(1)
    for name, M, (N0, K0), (N1, K1) in PAIRS:
        if args.only and not any(o in name for o in args.only.split(',')):
            continue
        t = {}
        ops = []
        for N, K in ((N0, K0), (N1, K1)):
            ops.append(((torch.rand(M, N, device=dev, generator=g) * 2 - 1).to(torch.bfloat16), (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16),
                        torch.empty(N, K, device=dev), torch.empty(N, device=dev)))
        scratch = torch.empty(lib.mae_linear_wgrad_pair_scratch_bytes(M, N0, K0, N1, K1), dtype=torch.uint8, device=dev)
        (y0, a0, w0, b0), (y1, a1, w1, b1) = ops

        def run(mode):
            os.environ["MAE_WGRAD_PAIR"] = mode
            check(lib.mae_linear_wgrad_pair(_ptr(y0), _ptr(a0), N0, K0, _ptr(w0), _ptr(b0), _ptr(y1), _ptr(a1), N1, K1, _ptr(w1), _ptr(b1), M, BF16, _ptr(scratch), stream(dev)))
        ref = None
        for mode in ("3", "4", "4b", "1", "0"):
            run(mode); torch.cuda.synchronize()
            outs = [x.clone() for x in (w0, b0, w1, b1)]
            if ref is None:
                ref = outs
            elif mode[0] == "4":
                same = all(torch.equal(a, b) for a, b in zip(ref, outs))
                print(f"  pair kernel tn{mode} bitwise equal to tn3: {same}", flush=True)
            ts = []
            for _ in range(args.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(mode); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            t[mode] = sorted(ts)[len(ts) // 2]
        fl = 2.0 * M * (N0 * K0 + N1 * K1)
        print(f"{name:16s} M={M} | paired (default): {t['1']:7.1f} us {fl / t['1'] / 1e6:5.0f} TF/s | tn3: {t['3']:7.1f} us | tn4: {t['4']:7.1f} us {fl / t['4'] / 1e6:5.0f} TF/s | tn4 burst: {t['4b']:7.1f} us | two launches: {t['0']:7.1f} us {fl / t['0'] / 1e6:5.0f} TF/s", flush=True)
    os.environ.pop("MAE_WGRAD_PAIR", None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="v2,v3")
    ap.add_argument("--only", default=None)
    ap.add_argument("--cold", action="store_true", help="evict the Infinity Cache before every timed launch (a 512 MiB fill), as inside a training step where the operand was written long before")
    ap.add_argument("--wgrad", action="store_true", help="time the weight-gradient GEMM (+ slab reduce) instead")
    ap.add_argument("--wgrad-pair", action="store_true", help="time the paired weight-gradient launch against the two launches it replaces")
    ap.add_argument("--wgrad-variants", default="v2", help="comma list of MAE_WGRAD values to A/B (v1 | v2 | v2r | v3 | v3r)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.wgrad_pair:
        return wgrad_pair_main(args, dev)
    if args.wgrad:
        return wgrad_main(args, dev)
    variants = args.variants.split(",")
    g = torch.Generator(device=dev).manual_seed(1)
    flush = torch.zeros(128 << 20, dtype=torch.int32, device=dev) if args.cold else None
    for name, M, N, K, epi, odt in SHAPES:
        if args.only and not any(o in name for o in args.only.split(',')):
            continue
        A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) / K ** 0.5).to(torch.bfloat16)
        bias = torch.rand(N, device=dev, generator=g)
        aux = None
        if epi == "resid":
            aux = torch.rand(M, N, device=dev, generator=g)
        elif epi in ("dgelu", "mul"):
            aux = (torch.rand(M, N, device=dev, generator=g) * 4 - 2).to(TDT[odt])
        outs, times = {}, {v: [] for v in variants}
        for v in variants:
            outs[v] = (torch.empty(M, N, dtype=TDT[odt], device=dev), torch.empty(M, N, dtype=TDT[odt], device=dev))

        def run(v):
            os.environ["MAE_GEMM_NT"] = v
            o, o2 = outs[v]
            check(lib.mae_linear_fwd(_ptr(A), _ptr(W), _ptr(bias), M, N, K, BF16, MODE[epi], odt, _ptr(o), _ptr(o2) if epi in ("gelu", "gelu_grad") else None,
                                     _ptr(aux) if aux is not None else None, stream(dev)))
        for v in variants:
            run(v)
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for v in variants:
                if args.cold:
                    flush.add_(1); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(v); e1.record(); torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3)
        osz = 4 if odt == F32 else 2
        nbytes = M * K * 2 + N * K * 2 + M * N * osz * (2 if epi in ("gelu", "gelu_grad") else 1) + (M * N * (4 if epi == "resid" else osz) if aux is not None else 0)
        line = f"{name:24s} M={M} N={N} K={K}"
        for v in variants:
            t = sorted(times[v])[len(times[v]) // 2]
            line += f" | {v}: {t:7.1f} us {2 * M * N * K / t / 1e6:6.0f} TF/s {nbytes / t / 1e3:5.0f} GB/s"
        if len(variants) == 2:
            same = torch.equal(outs[variants[0]][0], outs[variants[1]][0]) and (epi not in ("gelu", "gelu_grad") or torch.equal(outs[variants[0]][1], outs[variants[1]][1]))
            line += f" | bitwise-equal={same}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
