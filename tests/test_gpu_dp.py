"""The data-parallel PRODUCT step on the GPU (-m gpu): two gloo ranks sharing cuda:0, each a fresh child process running
MAEPretrainModule.fused_training_step over libmae_hip.so on its rows of the global batch (tests/dp_worker.py).

Asserted: both ranks end with identical parameters; they equal the single-process full-batch step; rank-sliced
noise gives the single-process masks bit for bit; the returned loss is the GLOBAL mean; the bucketed exchange that
overlaps the backward pass (gradient-ready events + async all-reduce on a side stream) is bit-identical to one blocking
all-reduce after the whole backward pass.  The reference has no multi-GPU path (devices=1,
scripts/training/pretrain_mae.py:118): the checker here is the single-process run of the same engine, itself checked
against the oracle in tests/test_gpu_engine.py."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(out, world, config, precision, batch, env_extra):
    port = _free_port()
    env = dict(os.environ, **env_extra)
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "dp_worker.py"), "--rank", str(r), "--world", str(world),
                               "--port", str(port), "--out", str(out), "--config", config, "--precision", precision,
                               "--global-batch", str(batch)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for p, o in zip(procs, logs):
        assert p.returncode == 0, f"rank failed ({p.returncode}):\n{o[-3000:]}"
    return [torch.load(out / f"w{world}_r{r}.pt", weights_only=True) for r in range(world)]


@pytest.mark.parametrize("config,precision,batch,tol", [("micro", "fp32", 8, 1e-5), ("vits8", "bf16", 8, 2e-4), ("ijepa", "fp32", 8, 2e-4)])
def test_two_rank_product_step_equals_single_process(tmp_path, config, precision, batch, tol):
    a, b, c = tmp_path / "overlap", tmp_path / "blocking", tmp_path / "single"
    for d in (a, b, c):
        d.mkdir()
    small = "8" if config == "vits8" else "0.01"   # several buckets even for the micro models
    ov = _run(a, 2, config, precision, batch, {"MAE_DP_OVERLAP": "1", "MAE_DP_BUCKET_MB": small})
    bl = _run(b, 2, config, precision, batch, {"MAE_DP_OVERLAP": "0"})
    one = _run(c, 1, config, precision, batch, {})[0]
    assert ov[0]["overlap"] and not bl[0]["overlap"]
    buckets = ov[0]["buckets"]
    n = buckets[0][2] - 1
    assert len(buckets) >= 3 and buckets[-1][1] == 0 and all(x[1] == y[2] for x, y in zip(buckets, buckets[1:]))  # tiles [0, n]
    assert buckets[0][2] == n + 1  # the first bucket carries the loss slot
    # every rank holds the same parameters after two steps; bucketed + overlapped == one blocking all-reduce, bit for bit
    assert torch.equal(ov[0]["params"], ov[1]["params"]) and torch.equal(bl[0]["params"], bl[1]["params"])
    assert torch.equal(ov[0]["params"], bl[0]["params"]) and torch.equal(ov[0]["losses"], bl[0]["losses"])
    assert torch.equal(ov[0]["stats"], bl[0]["stats"])
    # == the single-process full-batch step
    assert rel_err(ov[0]["params"], one["params"]) < tol
    if config == "ijepa":  # the EMA target follows the reduced update on every rank
        assert torch.equal(ov[0]["target"], ov[1]["target"]) and torch.equal(ov[0]["target"], bl[0]["target"])
        assert rel_err(ov[0]["target"], one["target"]) < tol
    assert not torch.equal(one["params"], torch.zeros_like(one["params"]))
    assert torch.equal(torch.cat([ov[0]["keep0"], ov[1]["keep0"]]), one["keep0"])          # masks bit-equal
    assert torch.equal(ov[0]["losses"], ov[1]["losses"])                                   # the global mean on every rank
    assert torch.allclose(ov[0]["losses"], one["losses"], rtol=(1e-6 if precision == "fp32" else 1e-4), atol=0)
    assert abs(float(ov[0]["stats"][0]) - float(one["stats"][0])) <= 10 * tol * float(one["stats"][0])  # global grad norm


@pytest.mark.parametrize("config,precision,world,batch,tol", [("micro", "fp32", 2, 7, 1e-5), ("micro", "fp32", 3, 2, 1e-5), ("ijepa", "fp32", 2, 5, 2e-4)])
def test_ragged_global_batch_equals_single_process(tmp_path, config, precision, world, batch, tol):
    """The last batch of an epoch (the reference never drops one, src/data.py:86-92): rows that do not divide by the world size are
    split 3 + 4, a 2-row batch over three ranks leaves rank 0 with nothing (it joins the same collectives with zeros).  Each rank
    weights its mean by rows_local / rows_global; the result equals the single-process step on the whole batch, overlapped and
    blocking exchange agree bit for bit, every rank reports the global mean loss."""
    a, b, c = tmp_path / "overlap", tmp_path / "blocking", tmp_path / "single"
    for d in (a, b, c):
        d.mkdir()
    ov = _run(a, world, config, precision, batch, {"MAE_DP_OVERLAP": "1", "MAE_DP_BUCKET_MB": "0.01"})
    bl = _run(b, world, config, precision, batch, {"MAE_DP_OVERLAP": "0"})
    one = _run(c, 1, config, precision, batch, {})[0]
    for r in range(1, world):
        assert torch.equal(ov[0]["params"], ov[r]["params"]) and torch.equal(ov[0]["losses"], ov[r]["losses"])
    assert torch.equal(ov[0]["params"], bl[0]["params"]) and torch.equal(ov[0]["losses"], bl[0]["losses"])
    assert rel_err(ov[0]["params"], one["params"]) < tol
    assert torch.allclose(ov[0]["losses"], one["losses"], rtol=1e-5, atol=0)
    assert torch.equal(torch.cat([o["keep0"] for o in ov]), one["keep0"])


@pytest.mark.parametrize("config,precision,world,batch,tol", [("micro", "fp32", 2, 8, 2e-6), ("micro", "fp32", 3, 7, 2e-6), ("vits8", "bf16", 2, 8, 2e-5)])
def test_sharded_optimizer_equals_the_replicated_step(tmp_path, config, precision, world, batch, tol):
    """MAE_DP_SHARDED_OPT=1: every rank receives its 1 / world slice of the summed gradient, adds its slice's sum of
    squares to the others', runs AdamW on the slice and all-gathers the parameters.  After two steps every rank holds the same
    parameters, equal to the replicated step's up to the order in which the squared gradients were summed (the clip coefficient may
    differ in its last bit), the loss is the same global mean and the reported gradient norm agrees."""
    a, b = tmp_path / "sharded", tmp_path / "replicated"
    for d in (a, b):
        d.mkdir()
    sh = _run(a, world, config, precision, batch, {"MAE_DP_SHARDED_OPT": "1"})
    rp = _run(b, world, config, precision, batch, {"MAE_DP_OVERLAP": "0"})
    for r in range(1, world):
        assert torch.equal(sh[0]["params"], sh[r]["params"]) and torch.equal(sh[0]["losses"], sh[r]["losses"])
    assert not torch.equal(sh[0]["params"], torch.zeros_like(sh[0]["params"]))
    assert rel_err(sh[0]["params"], rp[0]["params"]) < tol
    assert torch.allclose(sh[0]["losses"], rp[0]["losses"], rtol=1e-6 if precision == "fp32" else 1e-5, atol=0)
    assert abs(float(sh[0]["stats"][0]) - float(rp[0]["stats"][0])) <= 1e-5 * float(rp[0]["stats"][0])
    # gather_optimizer_state(): every rank then holds the whole AdamW state (what rank 0 writes into a checkpoint)
    for key in ("exp_avg", "exp_avg_sq"):
        assert torch.equal(sh[0][key], sh[world - 1][key]) and rel_err(sh[0][key], rp[0][key]) < 10 * tol
