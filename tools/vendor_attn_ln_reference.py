#!/usr/bin/env python3
"""Known-good references for the attention and LayerNorm kernels (methodology rule 10): torch's own GPU ops on the same data
(F.scaled_dot_product_attention forward + backward through autograd; F.layer_norm forward + backward), next to this
repository's kernels through the C ABI.  Measurement only: the product never calls torch for arithmetic.
    python tools/vendor_attn_ln_reference.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tests.util import BF16, check, lib, stream, _ptr  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)


def med(fn, n=7):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for name, B, T, H, hd in [("enc T=36 hd=64", 2000, 36, 6, 64), ("dec T=145 hd=32", 2000, 145, 6, 32), ("vit-b dec T=197 hd=32", 512, 197, 16, 32)]:
    qkv = torch.randn(B, T, 3, H, hd, device=dev, generator=g).to(torch.bfloat16)
    do = torch.randn(B, T, H * hd, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(B, T, H * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, T, device=dev)
    dqkv = torch.empty_like(qkv)
    t_f = med(lambda: check(lib.mae_attention_fwd(_ptr(qkv), B, T, H, hd, BF16, _ptr(out), _ptr(lse), stream(dev))))
    t_b = med(lambda: check(lib.mae_attention_bwd(_ptr(qkv), _ptr(out), _ptr(do), _ptr(lse), B, T, H, hd, BF16, _ptr(dqkv), stream(dev))))
    q, k, v = (qkv[:, :, i].transpose(1, 2).contiguous().requires_grad_(True) for i in range(3))  # (B, H, T, hd), as timm hands them to SDPA
    v_f = med(lambda: F.scaled_dot_product_attention(q, k, v))
    o = F.scaled_dot_product_attention(q, k, v)
    go = do.view(B, T, H, hd).transpose(1, 2).contiguous()
    v_b = med(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True))
    print(f"attention {name:22s} B={B}: ours fwd {t_f:7.1f} us bwd {t_b:7.1f} us | torch SDPA fwd {v_f:7.1f} us bwd {v_b:7.1f} us", flush=True)

for name, rows, dim in [("enc rows 72000 x 384", 72000, 384), ("dec rows 290000 x 192", 290000, 192), ("vit-b rows 25088 x 768", 25088, 768)]:
    x = torch.randn(rows, dim, device=dev, generator=g)
    br = torch.randn(rows, dim, device=dev, generator=g).to(torch.bfloat16)
    gam, bet = torch.rand(dim, device=dev, generator=g) + 0.5, torch.rand(dim, device=dev, generator=g)
    y = torch.empty(rows, dim, dtype=torch.bfloat16, device=dev)
    xo = torch.empty_like(x)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    t_f = med(lambda: check(lib.mae_add_layernorm_fwd(_ptr(x), _ptr(br), _ptr(xo), None, _ptr(gam), _ptr(bet), 1e-6, rows, dim, BF16, _ptr(y), _ptr(mean), _ptr(rstd), stream(dev))))
    dy = torch.randn(rows, dim, device=dev, generator=g).to(torch.bfloat16)
    dx = torch.zeros_like(x); dxc = torch.empty_like(y)
    dg, db = torch.empty(dim, device=dev), torch.empty(dim, device=dev)
    part = torch.empty(2 * 1024 * dim, device=dev)
    t_b = med(lambda: check(lib.mae_layernorm_bwd(_ptr(dy), BF16, _ptr(xo), None, _ptr(gam), _ptr(mean), _ptr(rstd), rows, dim, 1, _ptr(dx), _ptr(dxc), _ptr(dg), _ptr(db), _ptr(part), stream(dev))))
    # torch: the same work as separate ops (residual add in fp32, layer_norm under bf16 autocast semantics: fp32 math, bf16 out)
    xr = x.clone().requires_grad_(True)
    gr, br_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)

    def tf():
        s = xr + br.float()
        return s, F.layer_norm(s, (dim,), gr, br_, 1e-6).to(torch.bfloat16)
    v_f = med(tf)
    s, yy = tf()
    v_b = med(lambda: torch.autograd.grad(yy, (xr, gr, br_), dy, retain_graph=True))
    nb_f, nb_b = rows * dim * 12, rows * dim * 16
    print(f"add+layernorm {name:24s}: ours fwd {t_f:6.1f} us ({nb_f / t_f / 1e3:5.0f} GB/s) bwd {t_b:6.1f} us ({nb_b / t_b / 1e3:5.0f} GB/s) | torch fwd {v_f:6.1f} us bwd {v_b:6.1f} us", flush=True)
