#!/usr/bin/env python3
"""Headline benchmark: images/sec of one whole MAE pretrain step (zero_grad + mask + fwd + MSE + bwd + [grad all-reduce]
+ clip(1.0) + AdamW) on synthetic 96x96x3 batches, ViT-S/8 + the reference's YAML decoder (192 x 2 x 6 heads),
mask_ratio 0.75, bf16 MFMA operands / fp32 accumulate, batch 2000 per GPU (BASELINE.json configs[1]; SURVEY 8d "2a").

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One JSON line on rank 0.  `value` is whole-job images/s with inputs resident in HBM (K steps, barrier + synchronize on
both sides, max over ranks).  `roofline` prices the dominant kernel class (the bf16 MFMA Linear GEMM, forward + dgrad
launches) from HIP events recorded on the launch stream around every launch, in a second pass over the same K steps
(the ~860 events per step cost ~5 % and would otherwise distort `value`); `traffic` is the HBM bytes per launch of that
kernel from the committed rocprofv3 PMC passes.  `cpu_baseline` times the CPU oracle (the torch fp32 restatement of the
reference path) on the host cores, rank 0 / N=1 only, on a bounded sample (ViT-S/8, batch 64).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL; must be set before the GPU is initialised

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X (MI355X_MICROARCH.md chip table)
PEAK_HBM_GBS = 8000.0

GENERAL = dict(image_size=96, patch_size=8, in_chans=3, mask_ratio=0.75, engine_precision="bf16")
ENCODER = dict(embed_dim=384, depth=12, num_heads=6)
DECODER = dict(decoder_embed_dim=192, decoder_depth=2, decoder_num_heads=6)
TRAIN = dict(mask_ratio_start=0.75, mask_ratio_end=0.75, mask_ramp_epochs=5, total_epochs=800, warmup_epochs=20,
             batch_size=2000, base_learning_rate=1.5e-4, weight_decay=0.05)


def flops_per_image_step() -> float:
    """2*MAC, GEMMs + attention matmuls, patch-embed on visible patches only, step = 3 x fwd."""
    L, k, P, D, Dd, depth, dd = 145, 36, 192, 384, 192, 12, 2
    m = L - k
    fwd = (2 * (k - 1) * P * D + depth * k * 24 * D * D + depth * 4 * k * k * D + 2 * k * D * Dd
           + dd * L * 24 * Dd * Dd + dd * 4 * L * L * Dd + 2 * m * Dd * P)
    return 3.0 * fwd


def pmc_traffic(kernel: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    tools/pmc_traffic.py); PMC counters cannot be read from inside the timed process, so this is the profiled value."""
    f = ROOT / "profiles" / "r01_pmc_hbm_traffic_per_launch.json"
    try:
        return json.loads(f.read_text())[kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(batch: int = 64, steps: int = 3) -> dict:
    """The oracle (CPU restatement of the reference path) timed on this box's host cores: the checker used as a
    reported baseline, never as the product."""
    from oracle import mae_oracle as O
    torch.set_float32_matmul_precision("highest")
    cfg = O.VIT_S8_YAMLDEC
    params = O.init_params(cfg, 73)
    state = {}
    images = O.synthetic_images(batch, cfg)
    lr = O.effective_lr(1.5e-4, 2000) * O.lr_lambda(0, 20, 800)
    times = []
    for step in range(1, steps + 2):
        noise = O.make_noise(batch, cfg.sequence_length, torch.Generator().manual_seed(73 + step))
        t0 = time.perf_counter()
        O.train_step(params, cfg, state, images, noise, lr, step)
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": batch / t, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle.train_step fp32 'highest', ViT-S/8 + dec 192x2x6, batch {batch}, 1 warm-up + {steps} timed steps, median"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2000, help="images per GPU (weak scaling)")
    ap.add_argument("--no-kernel-timers", action="store_true", help="skip the second, event-instrumented pass (no roofline object)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="collective backend: nccl (= RCCL, the measured path) or gloo (rehearsal of the N > 1 control flow)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0 (with --backend gloo on a one-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            torch.distributed.init_process_group(args.backend)

    from ssrl_vit_mae_jepa_amd import MAEPretrainModule
    B = args.batch
    tcfg = dict(TRAIN, batch_size=B * world)
    torch.manual_seed(73)
    module = MAEPretrainModule(dict(general=GENERAL, encoder=ENCODER, decoder=DECODER), tcfg).to(dev)
    module.on_train_epoch_start()
    model = module.model
    L = model.sequence_length

    # synthetic inputs, resident in HBM before the timed region; every rank draws the global tensors and keeps its rows
    g = torch.Generator(device=dev).manual_seed(73)
    images = (torch.rand(B * world, 3, 96, 96, device=dev, generator=g) * 2 - 1)[rank * B:(rank + 1) * B].contiguous()
    total = args.warmup + args.steps
    noises = [torch.rand(B * world, L, device=dev, generator=g)[rank * B:(rank + 1) * B].contiguous() for _ in range(total)]

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        module.fused_training_step(images, noises[i])
    sync()
    t0 = time.perf_counter()
    loss = None
    for i in range(args.steps):
        loss = module.fused_training_step(images, noises[args.warmup + i])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # Per-kernel-class HIP events (two per launch, ~860 per step) cost ~5 % of the step, so `value` comes from the
    # un-instrumented region above and the roofline from a second pass over the SAME K steps with the events on.
    timers = not args.no_kernel_timers
    kernels, timed_ms = {}, None
    if timers:
        model.engine.timers_reset()
        model.engine.timers_enable(True)
        t1 = time.perf_counter()
        for i in range(args.steps):
            module.fused_training_step(images, noises[args.warmup + i])
        sync()
        timed_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        model.engine.timers_enable(False)
        kernels = model.engine.timers_read()
    if rank == 0:
        img_s = B * world * args.steps / elapsed
        fl = flops_per_image_step()
        out = {
            "metric": "images/sec pretrain step (fwd+bwd+opt), ViT-S/8 96px MAE, 1/2/4/8 MI355X",  # BASELINE.json's metric, verbatim
            "value": img_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"ViT-S/8 96px MAE (enc 384x12x6h, dec 192x2x6h, mask_ratio 0.75), batch {B}/GPU, full step",
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "algorithmic_gflop_per_image_step": fl / 1e9},
            "final_loss": float(loss.item()),
            "step_mfma_frac": img_s / world * fl / (PEAK_BF16_TFLOPS * 1e12),
        }
        if kernels:
            k = kernels["linear_nt"]
            ach = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt2_kernel (Linear fwd + dgrad, persistent LDS-DMA ring, bf16 MFMA 16x16x32)",
                               "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                               "traffic": pmc_traffic("gemm_nt2_kernel"), "avg_launch_us": 1e3 * k["ms"] / max(1, k["launches"]),
                               "launches": k["launches"], "algorithmic_bytes_per_launch": k["bytes"] / max(1, k["launches"]),
                               "measured_in": "second pass over the same K steps with per-launch HIP events on the launch stream",
                               "ms_per_step_with_events": timed_ms}
            tot_ms = sum(v["ms"] for v in kernels.values())
            out["kernels"] = {n: {"ms_per_step": v["ms"] / args.steps, "share": v["ms"] / tot_ms if tot_ms else 0.0,
                                  "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None,
                                  "gbs": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None}
                              for n, v in kernels.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
