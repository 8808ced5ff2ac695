#!/usr/bin/env python3
"""Headline benchmark: images/sec of one whole MAE pretrain step (zero_grad + mask + fwd + MSE + bwd + [grad all-reduce]
+ clip(1.0) + AdamW) on synthetic 96x96x3 batches, ViT-S/8 + the reference's YAML decoder (192 x 2 x 6 heads),
mask_ratio 0.75, bf16 MFMA operands / fp32 accumulate, batch 2000 per GPU (BASELINE.json configs[1]; SURVEY 8d "2a").

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --config configs/vitb16_dec512.yaml --batch 512      # another configs/*.yaml model; the default line is unchanged
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One JSON line on rank 0.  `value` is whole-job images/s with inputs resident in HBM (K steps, barrier + synchronize on
both sides, max over ranks).  `roofline` prices the dominant kernel class (the bf16 MFMA Linear GEMM, forward + dgrad
launches) from HIP events recorded on the launch stream around every launch, in a second pass over the same K steps
(the ~860 events per step cost ~5 % and would otherwise distort `value`): `bound` is the roof the class's flop per
algorithmic byte falls under (hbm below the 312 flop/B ridge), `frac` the fraction of that roof, `frac_mfma` and
`frac_hbm` both; `traffic` is the HBM bytes per launch of that kernel from the newest committed rocprofv3 PMC passes.  `cpu_baseline` times the CPU oracle (the torch fp32 restatement of the
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


def flops_per_image_step(general=GENERAL, encoder=ENCODER, decoder=DECODER) -> float:
    """2*MAC, GEMMs + attention matmuls, patch-embed on visible patches only, step = 3 x fwd."""
    N = (general["image_size"] // general["patch_size"]) ** 2
    L, P = N + 1, general["patch_size"] ** 2 * general.get("in_chans", 3)
    k = max(1, int(L * (1 - general.get("mask_ratio", 0.75))))
    D, depth, Dd, dd = encoder["embed_dim"], encoder["depth"], decoder["decoder_embed_dim"], decoder["decoder_depth"]
    m = L - k
    fwd = (2 * (k - 1) * P * D + depth * k * 24 * D * D + depth * 4 * k * k * D + 2 * k * D * Dd
           + dd * L * 24 * Dd * Dd + dd * 4 * L * L * Dd + 2 * m * Dd * P)
    return 3.0 * fwd


def jepa_flops_per_image_step(general, encoder, predictor, k: float, m: int) -> float:
    """I-JEPA: context encoder (k tokens) and predictor (nblk sequences of k + m tokens) forward + backward = 3 x forward,
    target encoder (all N patches, no gradient) 1 x forward; 2*MAC, GEMMs + attention matmuls."""
    N = (general["image_size"] // general["patch_size"]) ** 2
    P = general["patch_size"] ** 2 * general.get("in_chans", 3)
    D, depth, Dp, dp = encoder["embed_dim"], encoder["depth"], predictor["pred_embed_dim"], predictor["pred_depth"]
    nb = int(general.get("num_target_blocks", 4))
    enc = lambda t: 2 * t * P * D + depth * t * 24 * D * D + depth * 4 * t * t * D  # noqa: E731
    T = k + m
    pred = 2 * k * D * Dp + nb * (dp * T * 24 * Dp * Dp + dp * 4 * T * T * Dp + 2 * m * Dp * D)
    return 3.0 * (enc(k) + pred) + enc(N)


def _pmc_file():
    """The newest committed PMC reduction of the default workload (profiles/rNN_pmc_hbm_traffic_per_launch.json)."""
    files = sorted((ROOT / "profiles").glob("r[0-9][0-9]_pmc_hbm_traffic_per_launch.json"))
    return files[-1] if files else None


PMC_FILE = _pmc_file()
NT_KERNEL = "gemm_nt3_kernel"   # the dominant kernel class: Linear forward + dgrad launches (k_gemm_nt3.hip)


def pmc_traffic(kernel: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    tools/pmc_traffic.py); PMC counters cannot be read from inside the timed process, so this is the profiled value
    (null until a profile of this kernel is committed)."""
    try:
        return json.loads(PMC_FILE.read_text())[kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def hbm_budget(general, encoder, decoder, B: int, default_workload: bool = False) -> dict:
    """Algorithmic HBM bytes of one step under the precision policy (bf16 branch tensors, fp32 residual stream / statistics /
    gradients / optimizer state; every tensor written once and read by each consumer once), by tensor class, next to the
    measured PMC sum of the committed profile (default workload only).  Bytes per element of a (rows x width) block tensor:
      forward 72 = LN1 12 (x 4 + branch 2 -> x 4 + ln 2) + qkv 2+6 + attention 6+2 + proj 2+2 + LN2 12 + fc1 2+16 + fc2 8+2
      backward 120 = fc2 wgrad 10 + fc2 dgrad*slope 18 + fc1 wgrad 10 + fc1 dgrad 10 + LN2 16 + proj wgrad/dgrad 4+4
                     + attention 16 + qkv wgrad/dgrad 8+8 + LN1 16"""
    N = (general["image_size"] // general["patch_size"]) ** 2
    L, C, img = N + 1, general.get("in_chans", 3), general["image_size"]
    P = general["patch_size"] ** 2 * C
    k = max(1, int(L * (1 - general.get("mask_ratio", 0.75))))
    m = L - k
    D, depth, Dd, dd = encoder["embed_dim"], encoder["depth"], decoder["decoder_embed_dim"], decoder["decoder_depth"]
    Ue, Ud, Up = B * k * D, B * L * Dd, B * m
    per = {"mlp_hidden (act + slope + d_hidden)": 64, "layernorm passes over the fp32 residual (+ bf16 in/out)": 56, "qkv / d_qkv": 36,
           "attention out / proj / fc2 operands": 36}
    blocks = depth * Ue + dd * Ud
    by = {name: v * blocks for name, v in per.items()}
    n_train = (P * D + 2 * D) + depth * (12 * D * D + 13 * D) + 2 * D + Dd + D * Dd + Dd + dd * (12 * Dd * Dd + 13 * Dd) + 2 * Dd + P * Dd + P
    by["pixels, patch rows, token assembly, prediction head, loss"] = (
        2 * B * C * img * img * 4 + B * k * P * 2 * 3 + Ue * (4 + 12) * 2 + Ue * 6 + B * k * Dd * 2 * 2 + Ud * 4 * 2 + B * k * D * 2 * 3
        + Up * Dd * (2 + 16 + 2 * 3) + Up * P * (4 + 4 + 2 * 3))
    by["weights, gradients, AdamW state, operand copies"] = n_train * (4 + 4 + 30 + 6)  # wgrad write, norm read, AdamW sweep, transposed copy
    total = sum(by.values())
    out = {"algorithmic_bytes_per_step": total, "by_tensor_class": by, "hbm_floor_ms_at_5.5TBps": total / 5.5e12 * 1e3}
    try:
        pm = json.loads(PMC_FILE.read_text()) if default_workload else {}  # the committed PMC passes profiled the default workload
        if "_step" in pm:
            out["measured_pmc_bytes_per_step"] = pm["_step"]["hbm_bytes_per_step"]
            out["measured_over_algorithmic"] = pm["_step"]["hbm_bytes_per_step"] / total
            out["measured_on"] = pm["_step"].get("workload")
    except Exception:
        pass
    return out


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(general, encoder, decoder, label: str, batch: int = 64, steps: int = 3) -> dict:
    """The oracle (CPU restatement of the reference path) timed on this box's host cores: the checker used as a
    reported baseline, never as the product.  Two legs: fp32 matmuls at "highest" (the parity setting) and
    the reference's literal torch.set_float32_matmul_precision("medium") (scripts/utils.py:22)."""
    from oracle import mae_oracle as O
    cfg = O.MAEConfig.from_dicts(general, encoder, decoder)
    lr = O.effective_lr(1.5e-4, 2000) * O.lr_lambda(0, 20, 800)

    def leg(precision: str, n: int) -> float:
        torch.set_float32_matmul_precision(precision)
        params = O.init_params(cfg, 73)
        state = {}
        images = O.synthetic_images(batch, cfg)
        times = []
        for step in range(1, n + 2):
            noise = O.make_noise(batch, cfg.sequence_length, torch.Generator().manual_seed(73 + step))
            t0 = time.perf_counter()
            O.train_step(params, cfg, state, images, noise, lr, step, mask_ratio=general.get("mask_ratio", 0.75))
            times.append(time.perf_counter() - t0)
        return batch / sorted(times[1:])[len(times[1:]) // 2]

    hi = leg("highest", steps)
    med = leg("medium", max(1, steps - 1))
    torch.set_float32_matmul_precision("highest")
    return {"value": hi, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": cpu_model(),
            "value_matmul_precision_medium": med,
            "sample": f"oracle.train_step fp32, {label}, batch {batch}, 1 warm-up + {steps} timed steps ('highest') / "
                      f"{max(1, steps - 1)} ('medium', the reference's literal setting), median"}


def jepa_cpu_baseline(general, encoder, predictor, label: str, batch: int = 32, steps: int = 2) -> dict:
    """The I-JEPA oracle (our CPU restatement of DESIGN.md's specification; the reference has no I-JEPA code) timed on the host cores."""
    from oracle import jepa_oracle as J
    cfg = J.JEPAConfig(image_size=general["image_size"], patch_size=general["patch_size"], in_chans=general.get("in_chans", 3),
                       embed_dim=encoder["embed_dim"], depth=encoder["depth"], num_heads=encoder["num_heads"],
                       pred_embed_dim=predictor["pred_embed_dim"], pred_depth=predictor["pred_depth"], pred_num_heads=predictor["pred_num_heads"],
                       loss=general.get("loss", "mse"))
    torch.set_float32_matmul_precision("highest")
    p = J.init_params(cfg, 73)
    pt = {k: v.clone() for k, v in p.items()}
    state, times = {}, []
    images = J.M.synthetic_images(batch, cfg.as_mae())
    for step in range(1, steps + 2):
        ctx, tgt = J.sample_masks(cfg, batch, torch.Generator().manual_seed(step))
        t0 = time.perf_counter()
        J.train_step(p, pt, cfg, state, images, ctx, tgt, 1e-4, step, 0.996)
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": batch / t, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": cpu_model(),
            "sample": f"oracle.jepa train_step fp32 'highest', {label}, batch {batch}, 1 warm-up + {steps} timed steps, median"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (weak scaling); default 2000 (BASELINE configs[1])")
    ap.add_argument("--config", default=None, help="a configs/*.yaml model instead of the default ViT-S/8 + dec 192x2x6 workload")
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
    force_group = os.environ.get("MAE_DP_FORCE_EXCHANGE") == "1" and "MASTER_ADDR" in os.environ  # one-rank RCCL rehearsal under torchrun
    if world > 1 or force_group:
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            torch.distributed.init_process_group(args.backend)

    from ssrl_vit_mae_jepa_amd import MAEPretrainModule
    general, encoder, decoder, train = GENERAL, ENCODER, DECODER, TRAIN
    predictor = None
    if args.config:
        import yaml
        y = yaml.safe_load(open(args.config))
        predictor = y["model"].get("predictor")   # an I-JEPA config (configs/ijepa_*.yaml)
        encoder = y["model"]["encoder"]
        if predictor:
            general = dict(engine_precision="bf16", **y["model"]["general"])
            decoder, train = None, dict(y["pretrain"])
        else:
            general = dict(mask_ratio=float(y["pretrain"].get("mask_ratio_end", 0.75)), engine_precision="bf16", **y["model"]["general"])
            decoder = y["model"]["decoder"]
            train = dict(y["pretrain"], mask_ratio_start=general["mask_ratio"], mask_ratio_end=general["mask_ratio"])
    B = args.batch if args.batch else (2000 if not args.config else max(1, int(train["batch_size"]) // 8))
    img, chans = int(general["image_size"]), int(general.get("in_chans", 3))
    if predictor:
        label = (f"ViT {encoder['embed_dim']}x{encoder['depth']}x{encoder['num_heads']}h /{general['patch_size']} {img}px I-JEPA (no reference code), "
                 f"predictor {predictor['pred_embed_dim']}x{predictor['pred_depth']}x{predictor['pred_num_heads']}h, latent {general.get('loss', 'mse')}")
    else:
        label = (f"ViT {encoder['embed_dim']}x{encoder['depth']}x{encoder['num_heads']}h /{general['patch_size']} {img}px MAE, dec "
                 f"{decoder['decoder_embed_dim']}x{decoder['decoder_depth']}x{decoder['decoder_num_heads']}h")
    if not args.config:
        label = "ViT-S/8 96px MAE (enc 384x12x6h, dec 192x2x6h, mask_ratio 0.75)"
    tcfg = dict(train, batch_size=B * world)
    torch.manual_seed(73)
    if predictor:
        from ssrl_vit_mae_jepa_amd import IJEPAPretrainModule
        module = IJEPAPretrainModule(dict(general=general, encoder=encoder, predictor=predictor), tcfg).to(dev)
    else:
        module = MAEPretrainModule(dict(general=general, encoder=encoder, decoder=decoder), tcfg).to(dev)
    module.on_train_epoch_start()
    model = module.model
    L = (img // int(general["patch_size"])) ** 2 + 1

    # synthetic inputs, resident in HBM before the timed region; every rank draws ITS rows only (a generator seeded per rank: weak
    # scaling has no global tensor to agree on, and 8 ranks drawing 16 000 images each would cost 8 x the memory and the time)
    g = torch.Generator(device=dev).manual_seed(73 + 1009 * rank)
    images = torch.rand(B, chans, img, img, device=dev, generator=g) * 2 - 1
    total = args.warmup + args.steps
    if predictor:  # per-step multi-block masks of the GLOBAL batch (host sampler, seeded), this rank's rows, uploaded before timing
        mg = torch.Generator().manual_seed(73)   # the SAME block sizes on every rank (context length is a per-step scalar), own positions
        masks = [model.sample_masks(B, mg) for _ in range(total)]
        noises = [(c.to(dev), t.to(dev)) for c, t in masks]
        model.reserve_workspace(B, max(c.shape[1] for c, _t in masks), masks[0][1].shape[1], max(t.shape[2] for _c, t in masks))  # no reallocation inside the timed region
        step_fn = lambda i: module.fused_training_step(images, noises[i][0], noises[i][1])  # noqa: E731
    else:
        noises = [torch.rand(B, L, device=dev, generator=g) for _ in range(total)]
        step_fn = lambda i: module.fused_training_step(images, noises[i])  # noqa: E731

    def sync():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step_fn(i)
    sync()
    # one event per step on the launch stream (torch's current stream IS the engine's launch stream): p10/p50/p90
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    loss = None
    marks[0].record()
    for i in range(args.steps):
        loss = step_fn(args.warmup + i)
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(round(q * (len(step_ms) - 1))))]  # noqa: E731
    if torch.distributed.is_initialized():
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
            step_fn(args.warmup + i)
        sync()
        timed_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        model.engine.timers_enable(False)
        kernels = model.engine.timers_read()
    if rank == 0:
        img_s = B * world * args.steps / elapsed
        if predictor:
            ks = [noises[args.warmup + i][0].shape[1] for i in range(args.steps)]
            fl = jepa_flops_per_image_step(general, encoder, predictor, sum(ks) / len(ks), noises[args.warmup][1].shape[2])
        else:
            fl = flops_per_image_step(general, encoder, decoder)
        out = {
            # BASELINE.json's metric, verbatim, for the default workload; a --config run names its own model
            "metric": ("images/sec pretrain step (fwd+bwd+opt), ViT-S/8 96px MAE, 1/2/4/8 MI355X" if not args.config
                       else f"images/sec pretrain step (fwd+bwd+opt), {label}"),
            "value": img_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{label}, batch {B}/GPU, full step",
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "algorithmic_gflop_per_image_step": fl / 1e9},
            "final_loss": float(loss.item()),
            "step_ms_percentiles": {"p10": pct(0.1), "p50": pct(0.5), "p90": pct(0.9), "min": step_ms[0], "max": step_ms[-1],
                                    "measured_with": "one HIP event per step on the launch stream, inside the timed region"},
            "step_mfma_frac": img_s / world * fl / (PEAK_BF16_TFLOPS * 1e12),
        }
        if kernels:
            k = kernels["linear_nt"]
            sec = k["ms"] * 1e-3
            tf = k["flops"] / sec / 1e12 if sec > 0 else 0.0          # algorithmic 2*M*N*K of the class / its HIP-event time
            gbs = k["bytes"] / sec / 1e9 if sec > 0 else 0.0          # algorithmic operand + output + side-input bytes / the same time
            frac_mfma, frac_hbm = tf / PEAK_BF16_TFLOPS, gbs / PEAK_HBM_GBS
            # the roof that binds is the one the class's arithmetic intensity falls under: flop per byte against the ridge
            # peak_flops / peak_bytes (312 flop/B); `frac` is the fraction of THAT roof, both fractions are printed
            intensity = k["flops"] / k["bytes"] if k["bytes"] > 0 else float("inf")
            ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
            hbm_bound = intensity < ridge
            out["roofline"] = {"bound": "hbm" if hbm_bound else "mfma",
                               "kernel": NT_KERNEL + " (Linear fwd + dgrad, persistent LDS-DMA rings, bf16 MFMA 16x16x32)",
                               "achieved": gbs if hbm_bound else tf, "peak": PEAK_HBM_GBS if hbm_bound else PEAK_BF16_TFLOPS,
                               "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": frac_hbm if hbm_bound else frac_mfma,
                               "frac_mfma": frac_mfma, "frac_hbm": frac_hbm, "achieved_tflops": tf, "achieved_gbs": gbs,
                               "flop_per_byte": intensity, "ridge_flop_per_byte": ridge,
                               "traffic": pmc_traffic(NT_KERNEL), "avg_launch_us": 1e3 * k["ms"] / max(1, k["launches"]),
                               "launches": k["launches"], "algorithmic_bytes_per_launch": k["bytes"] / max(1, k["launches"]),
                               "algorithmic_flops_per_launch": k["flops"] / max(1, k["launches"]),
                               "measured_in": "second pass over the same K steps with per-launch HIP events on the launch stream",
                               "ms_per_step_with_events": timed_ms}
            tot_ms = sum(v["ms"] for v in kernels.values())
            out["kernels"] = {n: {"ms_per_step": v["ms"] / args.steps, "share": v["ms"] / tot_ms if tot_ms else 0.0,
                                  "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None,
                                  "gbs": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None}
                              for n, v in kernels.items()}
        if not predictor:
            out["hbm_budget"] = hbm_budget(general, encoder, decoder, B, default_workload=(not args.config and B == 2000))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = (jepa_cpu_baseline(general, encoder, predictor, label, batch=32 if img <= 96 else 4) if predictor
                                   else cpu_baseline(general, encoder, decoder, label, batch=64 if img <= 96 else 8))
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
