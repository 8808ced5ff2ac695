"""World-size-2 data-parallel logic on CPU (gloo): rank-sliced images and noise, grad_scale = 1/world, ONE all-reduce
of the flat gradient arena, identical clip + AdamW on every rank == the single-process full-batch step.  The compute
stand-in here is the CPU oracle (no GPU in this container); on the GPU box the same functions wrap the native step."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    torch.set_float32_matmul_precision("highest")
    from oracle import mae_oracle as O
    from ssrl_vit_mae_jepa_amd import dist as mdist
    r, w = mdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = O.MAEConfig(image_size=32, patch_size=8, in_chans=3, embed_dim=48, depth=1, num_heads=2, decoder_embed_dim=64,
                      decoder_depth=1, decoder_num_heads=2)
    params = O.init_params(cfg, 73); O.randomize_params(params)
    B = 8
    images = O.synthetic_images(B, cfg)
    noise = mdist.global_noise(B, cfg.sequence_length, 73, 0, torch.device("cpu"))
    my_images, my_noise = mdist.shard_rows(images, rank, world), mdist.shard_rows(noise, rank, world)
    loss, grads, aux = O.loss_and_grads(params, cfg, my_images, my_noise)
    names = list(grads)
    flat = torch.cat([g.reshape(-1) for g in grads.values()]) / world  # grad_scale = 1/world
    mdist.allreduce_sum_(flat)                                        # the single collective of the step
    # replicated clip + AdamW
    off, g2 = 0, {}
    for n in names:
        g2[n] = flat[off:off + grads[n].numel()].view_as(grads[n]).clone(); off += grads[n].numel()
    total, _ = O.clip_grad_norm(g2, 1.0)
    O.adamw_step(params, g2, {}, 1e-3, 1)
    torch.save({"flat": flat, "total": total, "keep": aux["idx_keep"], "p": params["decoder.decoder_pred.weight"]}, f"{out_dir}/r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_step(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle import mae_oracle as O
    from ssrl_vit_mae_jepa_amd import dist as mdist
    torch.set_float32_matmul_precision("highest")
    cfg = O.MAEConfig(image_size=32, patch_size=8, in_chans=3, embed_dim=48, depth=1, num_heads=2, decoder_embed_dim=64,
                      decoder_depth=1, decoder_num_heads=2)
    params = O.init_params(cfg, 73); O.randomize_params(params)
    images = O.synthetic_images(8, cfg)
    noise = mdist.global_noise(8, cfg.sequence_length, 73, 0, torch.device("cpu"))
    loss, grads, aux = O.loss_and_grads(params, cfg, images, noise)
    flat = torch.cat([g.reshape(-1) for g in grads.values()])
    total, _ = O.clip_grad_norm(grads, 1.0)
    O.adamw_step(params, grads, {}, 1e-3, 1)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["flat"], r1["flat"])                                   # every rank holds the same reduced grads
    assert torch.allclose(r0["flat"], flat, rtol=1e-4, atol=1e-7)                # == full-batch gradient (mean of means)
    assert abs(float(r0["total"]) - float(total)) < 1e-4 * float(total)          # global norm, not per-rank norm
    assert torch.equal(torch.cat([r0["keep"], r1["keep"]]), aux["idx_keep"])     # rank-sliced noise -> identical masks
    assert torch.allclose(r0["p"], params["decoder.decoder_pred.weight"], rtol=1e-4, atol=1e-6) and torch.equal(r0["p"], r1["p"])


def _bucket_worker(rank, world, port, out_dir):
    """The product's bucket plan (MAEPretrainModule.gradient_buckets, host logic over the engine's gradient-ready points)
    driven over gloo on a CPU buffer: bucketed async all-reduces in backward order == one all-reduce of the whole buffer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MAE_DP_BUCKET_MB="0.05")
    torch.set_num_threads(2)
    from ssrl_vit_mae_jepa_amd import MAEPretrainModule
    from ssrl_vit_mae_jepa_amd import dist as mdist
    mdist.init_from_env(backend="gloo")
    module = MAEPretrainModule(dict(general=dict(image_size=32, patch_size=8), encoder=dict(embed_dim=48, depth=4, num_heads=2),
                                    decoder=dict(decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2)), {})
    n = module.model.engine.trainable_elems
    buckets = module.gradient_buckets()
    g = torch.Generator().manual_seed(1000 + rank)
    buf = torch.randn(n + module.model.GRAD_TAIL, generator=g)
    whole = buf.clone()
    works = [dist.all_reduce(buf[b:e], async_op=True) for _j, b, e in buckets]
    for w in works:
        w.wait()
    dist.all_reduce(whole[:n + 1])
    torch.save({"bucketed": buf, "whole": whole, "buckets": buckets, "n": n}, f"{out_dir}/b{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_plan_tiles_the_arena_and_equals_one_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "b0.pt"), torch.load(tmp_path / "b1.pt")
    n, buckets = r0["n"], r0["buckets"]
    assert buckets == r1["buckets"] and len(buckets) >= 3
    assert buckets[0][2] == n + 1 and buckets[-1][1] == 0                           # loss slot in the first bucket; ends at 0
    assert all(x[1] == y[2] for x, y in zip(buckets, buckets[1:]))                  # contiguous, in backward order
    assert [x[0] for x in buckets] == sorted(x[0] for x in buckets)                 # ready points are reached in this order
    assert torch.equal(r0["bucketed"][:n + 1], r0["whole"][:n + 1]) and torch.equal(r0["bucketed"][:n + 1], r1["bucketed"][:n + 1])


def test_gradient_ready_points_follow_the_arena_layout():
    from ssrl_vit_mae_jepa_amd import MaskedAutoencoder
    model = MaskedAutoencoder(dict(image_size=32, patch_size=8), dict(embed_dim=48, depth=3, num_heads=2),
                              dict(decoder_embed_dim=64, decoder_depth=2, decoder_num_heads=2))
    off = {name: o for name, o, _n, _s, _f in model.engine.table}
    pts = model.grad_ready_points()
    assert pts == [off["decoder.mask_token"], off["encoder.vit.blocks.2.norm1.weight"], off["encoder.vit.blocks.1.norm1.weight"], 0]
    # everything the decoder owns lies behind point 0, the encoder's final norm right in front of it
    assert all(off[n] >= pts[0] for n in off if n.startswith("decoder.") and n != "decoder.decoder_pos_embed")
    assert pts[1] < off["encoder.vit.norm.weight"] < pts[0]


def test_shard_bounds_tile_any_batch():
    """Contiguous, as-even-as-possible row shards: they tile the batch for every (rows, world), sizes differ by at most one, and a
    rank may own nothing when the batch is shorter than the world (its step then contributes zeros to the same collectives)."""
    from ssrl_vit_mae_jepa_amd import dist as mdist
    for n in (0, 1, 2, 7, 8, 9, 10, 2000, 94000 % 2000):
        for w in (1, 2, 3, 8):
            b = [mdist.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    t = torch.arange(14).reshape(7, 2)
    assert torch.equal(torch.cat([mdist.shard_rows(t, r, 2) for r in range(2)]), t)
    assert mdist.shard_rows(t, 0, 2).shape[0] == 3 and mdist.shard_rows(t, 1, 2).shape[0] == 4


def _ragged_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    torch.set_float32_matmul_precision("highest")
    from oracle import mae_oracle as O
    from ssrl_vit_mae_jepa_amd import dist as mdist
    mdist.init_from_env(backend="gloo")
    cfg = O.MAEConfig(image_size=32, patch_size=8, in_chans=3, embed_dim=48, depth=1, num_heads=2, decoder_embed_dim=64,
                      decoder_depth=1, decoder_num_heads=2)
    params = O.init_params(cfg, 73); O.randomize_params(params)
    B = 7   # ragged over two ranks (3 + 4 rows); over three ranks of a 2-row batch rank 0 would own nothing
    images = O.synthetic_images(B, cfg)
    noise = mdist.global_noise(B, cfg.sequence_length, 73, 0, torch.device("cpu"))
    lo, hi = mdist.shard_bounds(B, rank, world)
    names = O.trainable_names(cfg)
    n = sum(params[k].numel() for k in names)
    flat = torch.zeros(n + 1)
    if hi > lo:
        loss, grads, _aux = O.loss_and_grads(params, cfg, images[lo:hi], noise[lo:hi])
        w = (hi - lo) / B                                   # the rank's share of the global batch: what fused_training_step passes
        flat[:n] = torch.cat([grads[k].reshape(-1) for k in names]) * w
        flat[n] = loss * w                                  # the loss slot behind the gradients
    mdist.allreduce_sum_(flat)
    torch.save(flat, f"{out_dir}/ragged_r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_batch_weighted_by_rows_equals_the_single_process_step(tmp_path):
    """The last batch of an epoch need not divide by the world size (the reference never drops a batch, src/data.py:86-92): each
    rank scales its mean loss and gradient by rows_local / rows_global before the one sum, which is then the gradient and the
    loss of the whole batch.  CPU stand-in for MAEPretrainModule._exchanged_loss_and_grads(weight=...)."""
    world, port = 2, _free_port()
    mp.spawn(_ragged_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle import mae_oracle as O
    from ssrl_vit_mae_jepa_amd import dist as mdist
    torch.set_float32_matmul_precision("highest")
    cfg = O.MAEConfig(image_size=32, patch_size=8, in_chans=3, embed_dim=48, depth=1, num_heads=2, decoder_embed_dim=64,
                      decoder_depth=1, decoder_num_heads=2)
    params = O.init_params(cfg, 73); O.randomize_params(params)
    images = O.synthetic_images(7, cfg)
    noise = mdist.global_noise(7, cfg.sequence_length, 73, 0, torch.device("cpu"))
    loss, grads, _ = O.loss_and_grads(params, cfg, images, noise)
    want = torch.cat([torch.cat([grads[k].reshape(-1) for k in O.trainable_names(cfg)]), loss.reshape(1)])
    r0, r1 = torch.load(tmp_path / "ragged_r0.pt"), torch.load(tmp_path / "ragged_r1.pt")
    assert torch.equal(r0, r1)
    assert float((r0 - want).norm() / want.norm()) < 1e-6
