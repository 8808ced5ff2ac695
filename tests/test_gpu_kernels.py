"""Kernel-level parity (-m gpu): every exported HIP kernel, called through the C ABI, against a plain fp32 torch
restatement of the same op on the same seeded inputs.  Integer outputs bit-exact; fp32 kernels to ~1e-5; bf16 kernels
against an fp32 reference fed the same bf16-rounded operands, tolerance stated per test."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import BF16, F32, TDT, check, dv, lib, max_err, rel_err, release_device_copies, stream, _ptr

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _release():
    yield
    release_device_copies()


def G(seed):
    return torch.Generator().manual_seed(seed)


# ----------------------------------------------------------------------------------------------- mask
@pytest.mark.parametrize("B,L,k", [(1, 145, 36), (7, 145, 36), (64, 145, 72), (3, 17, 4), (2, 145, 145), (5, 257, 1), (2000, 145, 36)])
def test_mask_from_noise_bit_exact(dev, B, L, k):
    noise = torch.rand(B, L, generator=G(73 + B))
    ref = noise.clone()
    ref[:, 0] = -1
    order = torch.argsort(ref, dim=1, stable=True)
    keep = torch.empty(B, k, dtype=torch.int64, device=dev)
    mask = torch.empty(B, max(L - k, 1), dtype=torch.int64, device=dev)
    check(lib.mae_mask_from_noise(_ptr(dv(noise)), B, L, k, _ptr(keep), _ptr(mask), stream(dev)))
    assert torch.equal(keep.cpu(), order[:, :k])
    if L > k:
        assert torch.equal(mask.cpu()[:, : L - k], order[:, k:])
    assert bool((keep[:, 0] == 0).all())


def test_mask_ties_broken_by_index(dev):
    """Equal noise values: the HIP kernel orders them by index (== torch.argsort(stable=True)).  lightly calls the
    UNSTABLE torch.argsort, whose order inside a run of equal keys is implementation-defined on CPU (AVX-512 builds of
    torch >= 2.3 reverse about half of such pairs -- measured here), so for tied rows the contract is equality up to the
    order inside each run of equal keys; tie-free rows are bit-exact (tests above, p(tie) ~ 6e-4 per 145-token row)."""
    noise = torch.rand(4, 145, generator=G(5))
    noise[0, 10] = noise[0, 99]
    noise[1, 3:9] = 0.25
    noise[2, :] = 0.5
    noise[3, 144] = noise[3, 0]
    ref = noise.clone(); ref[:, 0] = -1
    order = torch.argsort(ref, dim=1, stable=True)
    keep = torch.empty(4, 36, dtype=torch.int64, device=dev)
    mask = torch.empty(4, 109, dtype=torch.int64, device=dev)
    check(lib.mae_mask_from_noise(_ptr(dv(noise)), 4, 145, 36, _ptr(keep), _ptr(mask), stream(dev)))
    got = torch.cat([keep, mask], 1).cpu()
    assert torch.equal(got, order)
    assert keep[2].tolist() == list(range(36))
    unstable = torch.argsort(ref, dim=1)  # what the reference would produce on this host
    assert torch.equal(torch.gather(ref, 1, unstable), torch.gather(ref, 1, got))  # same keys position by position
    assert torch.equal(unstable.sort(1).values, got.sort(1).values)               # both are permutations


# ----------------------------------------------------------------------------------------------- patchify / mse
@pytest.mark.parametrize("B,C,img,p,m", [(3, 3, 96, 8, 109), (2, 3, 32, 8, 12), (2, 1, 28, 7, 5), (2, 3, 224, 14, 100)])
def test_patchify_gather(dev, B, C, img, p, m):
    g = G(1)
    images = torch.rand(B, C, img, img, generator=g) * 2 - 1
    n = (img // p) ** 2
    idx = torch.stack([torch.randperm(n + 1, generator=g)[:m] for _ in range(B)])
    patches = images.reshape(B, C, img // p, p, img // p, p)
    patches = torch.einsum("nchpwq->nhwpqc", patches).reshape(B, n, p * p * C)
    ref = torch.gather(patches, 1, (idx - 1).clamp(min=0).unsqueeze(-1).expand(-1, -1, p * p * C))
    out = torch.empty(B, m, p * p * C, device=dev)
    check(lib.mae_patchify_gather(_ptr(dv(images)), F32, _ptr(dv(idx)), B, C, img, p, m, _ptr(out), stream(dev)))
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("n", [4, 192 * 109 * 3, 4096 * 1024 + 8])
def test_mse_loss(dev, n):
    g = G(2)
    a, b = torch.randn(n, generator=g), torch.randn(n, generator=g)
    loss = torch.zeros(1, device=dev); dp = torch.empty(n, device=dev); scratch = torch.zeros(8192, device=dev)
    check(lib.mae_mse_loss(_ptr(dv(a)), _ptr(dv(b)), n, 0.5, _ptr(loss), _ptr(dp), _ptr(scratch), stream(dev)))
    ref = F.mse_loss(a.double(), b.double())
    assert abs(loss.item() - ref.item()) <= 2e-6 * ref.item()
    assert rel_err(dp, 0.5 * 2 * (a - b) / n) < 1e-6


# ----------------------------------------------------------------------------------------------- layernorm
@pytest.mark.parametrize("rows,dim", [(5, 144), (72, 192), (1000, 384), (33, 768), (9, 1024), (4, 8), (1001, 384), (7, 48), (70001, 192)])  # 16 / 32 / 64 lanes per row, row counts that leave a lane group without a row
@pytest.mark.parametrize("dt", [F32, BF16])
def test_layernorm_fwd_bwd(dev, rows, dim, dt):
    g = G(rows + dim)
    x = torch.randn(rows, dim, generator=g) * 2 + 0.3
    gamma, beta = torch.randn(dim, generator=g), torch.randn(dim, generator=g)
    dy = torch.randn(rows, dim, generator=g)
    res = torch.randn(rows, dim, generator=g)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    yref = F.layer_norm(xr, (dim,), gr, br, 1e-6)
    dy_used = dy.to(TDT[dt]).float()
    yref.backward(dy_used)
    y = torch.empty(rows, dim, dtype=TDT[dt], device=dev)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    check(lib.mae_layernorm_fwd(_ptr(dv(x)), None, _ptr(dv(gamma)), _ptr(dv(beta)), 1e-6, rows, dim, dt, _ptr(y),
                                _ptr(mean), _ptr(rstd), stream(dev)))
    tol = 1e-5 if dt == F32 else 6e-3
    assert rel_err(y.float(), yref) < tol
    assert rel_err(mean, x.mean(1)) < 1e-5 and rel_err(rstd, (x.var(1, unbiased=False) + 1e-6).rsqrt()) < 1e-5
    dx = res.to(dev).clone(); dxc = torch.empty(rows, dim, dtype=TDT[dt], device=dev)
    dg = torch.empty(dim, device=dev); db = torch.empty(dim, device=dev)
    partial = torch.zeros(2 * 1024 * dim, device=dev)
    check(lib.mae_layernorm_bwd(_ptr(dv(dy, TDT[dt])), dt, _ptr(dv(x)), None, _ptr(dv(gamma)), _ptr(mean), _ptr(rstd),
                                rows, dim, 1, _ptr(dx), _ptr(dxc), _ptr(dg), _ptr(db), _ptr(partial), stream(dev)))
    assert rel_err(dx, res + xr.grad) < 2e-5
    assert rel_err(dxc.float(), res + xr.grad) < tol
    assert rel_err(dg, gr.grad) < 2e-5 and rel_err(db, br.grad) < 2e-5


def test_layernorm_row_map(dev):
    g = G(9)
    rows_src, dim, n = 50, 192, 17
    x = torch.randn(rows_src, dim, generator=g)
    gamma, beta = torch.randn(dim, generator=g), torch.randn(dim, generator=g)
    rmap = torch.randperm(rows_src, generator=g)[:n].to(torch.int32)
    y = torch.empty(n, dim, device=dev); mean = torch.empty(n, device=dev); rstd = torch.empty(n, device=dev)
    check(lib.mae_layernorm_fwd(_ptr(dv(x)), _ptr(dv(rmap)), _ptr(dv(gamma)), _ptr(dv(beta)), 1e-6, n, dim, F32,
                                _ptr(y), _ptr(mean), _ptr(rstd), stream(dev)))
    ref = F.layer_norm(x[rmap.long()], (dim,), gamma, beta, 1e-6)
    assert rel_err(y, ref) < 1e-5
    dy = torch.randn(n, dim, generator=g)
    xr = x.clone().requires_grad_(True)
    F.layer_norm(xr[rmap.long()], (dim,), gamma, beta, 1e-6).backward(dy)
    dx = torch.zeros(rows_src, dim, device=dev); dg = torch.empty(dim, device=dev); db = torch.empty(dim, device=dev)
    partial = torch.zeros(2 * 1024 * dim, device=dev)
    check(lib.mae_layernorm_bwd(_ptr(dv(dy)), F32, _ptr(dv(x)), _ptr(dv(rmap)), _ptr(dv(gamma)), _ptr(mean), _ptr(rstd),
                                n, dim, 0, _ptr(dx), None, _ptr(dg), _ptr(db), _ptr(partial), stream(dev)))
    assert rel_err(dx, xr.grad) < 2e-5


@pytest.mark.parametrize("dt", [F32, BF16])
def test_add_layernorm_fwd(dev, dt):
    g = G(11)
    rows, dim, n = 40, 384, 23
    x = torch.randn(rows, dim, generator=g); br = torch.randn(rows, dim, generator=g).to(TDT[dt])
    gamma, beta = torch.randn(dim, generator=g), torch.randn(dim, generator=g)
    rmap = torch.randperm(rows, generator=g)[:n].to(torch.int32)
    for use_map in (False, True):
        nr = n if use_map else rows
        xo = torch.full((rows, dim), 7.0, device=dev)
        y = torch.empty(nr, dim, dtype=TDT[dt], device=dev); mean = torch.empty(nr, device=dev); rstd = torch.empty(nr, device=dev)
        check(lib.mae_add_layernorm_fwd(_ptr(dv(x)), _ptr(dv(br)), _ptr(xo), _ptr(dv(rmap)) if use_map else None, _ptr(dv(gamma)),
                                        _ptr(dv(beta)), 1e-6, nr, dim, dt, _ptr(y), _ptr(mean), _ptr(rstd), stream(dev)))
        v = x + br.float()
        sel = rmap.long() if use_map else torch.arange(rows)
        assert torch.equal(xo.cpu()[sel], v[sel])
        if use_map:
            rest = torch.ones(rows, dtype=torch.bool); rest[sel] = False
            assert bool((xo.cpu()[rest] == 7.0).all())  # rows outside the map are untouched
        assert rel_err(y.float(), F.layer_norm(v[sel], (dim,), gamma, beta, 1e-6)) < (1e-5 if dt == F32 else 6e-3)


# ----------------------------------------------------------------------------------------------- linear
def _gelu_grad(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


LIN_SHAPES = [(70, 144, 192), (72, 432, 144), (300, 576, 144), (145, 192, 768), (513, 384, 384), (1000, 1152, 384),
              (257, 1536, 384), (130, 384, 1536), (36, 192, 384), (218, 192, 192), (64, 16, 32), (31, 20, 24),
              (1300, 432, 144), (1300, 144, 576)]  # the last two: ragged MFMA tiles (N, K multiples of 8 only), several row tiles


@pytest.mark.parametrize("M,N,K", LIN_SHAPES)
@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("epi", ["none", "gelu", "resid", "dgelu", "none_f32out", "gelu_grad", "mul", "gelu_act"])
def test_linear_fwd(dev, M, N, K, dt, epi):
    if dt == F32 and epi == "none_f32out":
        pytest.skip("same as none")
    g = G(M * 7 + N)
    A = (torch.randn(M, K, generator=g)).to(TDT[dt]); W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(TDT[dt])
    bias = torch.randn(N, generator=g)
    acc = A.double() @ W.double().t()
    odt = F32 if (epi in ("resid", "none_f32out") or dt == F32) else BF16
    out = torch.full((M, N), float("nan"), dtype=TDT[odt], device=dev)
    out2 = torch.full((M, N), float("nan"), dtype=TDT[odt], device=dev)
    aux = None
    mode = {"none": 0, "none_f32out": 0, "gelu": 1, "resid": 2, "dgelu": 3, "gelu_grad": 4, "mul": 5, "gelu_act": 6}[epi]
    if epi == "resid":
        aux = torch.randn(M, N, generator=g)
        ref = aux.double() + acc + bias.double()
    elif epi == "dgelu":
        aux = torch.randn(M, N, generator=g).to(TDT[odt])
        ref = (acc + bias.double()) * _gelu_grad(aux.double())
    elif epi == "mul":
        aux = torch.randn(M, N, generator=g).to(TDT[odt])
        ref = (acc + bias.double()) * aux.double()
    elif epi == "gelu_grad":
        ref = _gelu_grad((acc + bias.double()).to(TDT[odt]).double())
    elif epi == "gelu_act":  # value only (forward-only passes): gelu of the pre-activation as it would have been stored
        ref = F.gelu((acc + bias.double()).to(TDT[odt]).double())
    else:
        ref = acc + bias.double()
    check(lib.mae_linear_fwd(_ptr(dv(A)), _ptr(dv(W)), _ptr(dv(bias)), M, N, K, dt, mode, odt, _ptr(out),
                             _ptr(out2) if epi in ("gelu", "gelu_grad") else None, _ptr(dv(aux)) if aux is not None else None, stream(dev)))
    tol = 2e-5 if odt == F32 and dt == F32 else (1e-5 if odt == F32 else 5e-3)
    assert rel_err(out.float(), ref) < tol
    if epi == "gelu":
        assert rel_err(out2.float(), F.gelu(out.float().cpu().double())) < (1e-5 if odt == F32 else 5e-3)
    if epi == "gelu_grad":
        assert rel_err(out2.float(), F.gelu((acc + bias.double()).to(TDT[odt]).double())) < (2e-5 if odt == F32 else 5e-3)


# The deferred-epilogue kernel retires a tile's GELU epilogue inside the next tile's K-loop: it needs workgroups that own
# several tiles (more than 256 tiles), a ragged last row of tiles, and every units-per-step schedule (K = 192 / 384 / 768).
@pytest.mark.parametrize("M,N,K,has_bias", [(18020, 1536, 384, True), (40100, 768, 192, True), (17000, 512, 768, False), (66000, 128, 192, True)])
def test_linear_fwd_gelu_grad_many_tiles(dev, M, N, K, has_bias):
    g = G(M + N + K)
    A = (torch.randn(M, K, generator=g)).to(torch.bfloat16); W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, generator=g) if has_bias else None
    Ad, Wd = A.to(dev), W.to(dev)
    pre = (Ad.float() @ Wd.float().t()) + (bias.to(dev) if has_bias else 0.0)   # fp32 products of bf16 values, fp32 sum
    pre = pre.to(torch.bfloat16).double()
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    out2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    check(lib.mae_linear_fwd(_ptr(Ad), _ptr(Wd), _ptr(dv(bias)) if has_bias else None, M, N, K, BF16, 4, BF16, _ptr(out), _ptr(out2), None, stream(dev)))
    torch.cuda.synchronize()
    assert not torch.isnan(out.float()).any() and not torch.isnan(out2.float()).any()
    # the reference pre-activation is rounded to bf16 from a differently-ordered fp32 sum: compare with a tolerance of a
    # few bf16 ulps on the slope / value, and tightly in the mean
    slope_ref, act_ref = _gelu_grad(pre), F.gelu(pre)
    assert rel_err(out.double(), slope_ref) < 5e-3 and rel_err(out2.double(), act_ref) < 5e-3
    assert (out.double() - slope_ref).abs().max() < 0.05 and (out2.double() - act_ref).abs().max() < 0.08


@pytest.mark.parametrize("M,N,K", [(70, 144, 192), (5000, 432, 144), (4100, 192, 768), (9000, 384, 384), (300, 16, 32), (8200, 1536, 384), (6000, 384, 1536), (4289, 576, 192), (20000, 1152, 384), (4096, 192, 192),
                                   (5000, 512, 1024), (4500, 1024, 512), (4200, 1536, 512), (4100, 328, 200)])  # + widths with a partial last 192-tile column
@pytest.mark.parametrize("dt", [F32, BF16])
def test_linear_wgrad(dev, M, N, K, dt):
    g = G(M + N + K)
    dY = (torch.randn(M, N, generator=g)).to(TDT[dt]); A = (torch.randn(M, K, generator=g)).to(TDT[dt])
    ref = dY.double().t() @ A.double()
    dW = torch.full((N, K), float("nan"), device=dev); db = torch.full((N,), float("nan"), device=dev)
    scratch = torch.zeros(lib.mae_linear_wgrad_scratch_bytes(M, N, K), dtype=torch.uint8, device=dev)
    check(lib.mae_linear_wgrad(_ptr(dv(dY)), _ptr(dv(A)), M, N, K, dt, _ptr(dW), _ptr(db), _ptr(scratch), stream(dev)))
    assert rel_err(dW, ref) < 2e-5
    assert rel_err(db, dY.double().sum(0)) < 2e-5


@pytest.mark.parametrize("M,s0,s1", [(9000, (384, 1536), (1536, 384)), (8300, (384, 384), (1152, 384)), (8192, (192, 192), (576, 192)),
                                     (9001, (512, 1024), (1536, 512)), (5000, (384, 384), (1152, 384)), (8500, (144, 144), (432, 144))])
@pytest.mark.parametrize("dt", [F32, BF16])
def test_linear_wgrad_pair(dev, M, s0, s1, dt):
    """Two weight gradients over the same rows in one launch (a block's fc2 + fc1, proj + qkv) against fp64; the last three
    cases take the fallback to two ordinary launches (ragged tile columns are fine, M < 8192 and 144-wide layers are not)."""
    g = G(M + s0[0] + s1[1])
    outs = []
    args = []
    for N, K in (s0, s1):
        dY = torch.randn(M, N, generator=g).to(TDT[dt]); A = torch.randn(M, K, generator=g).to(TDT[dt])
        dW = torch.full((N, K), float("nan"), device=dev); db = torch.full((N,), float("nan"), device=dev)
        outs.append((dW, db, dY.double().t() @ A.double(), dY.double().sum(0)))
        args.append((dv(dY), dv(A), N, K, dW, db))
    nbytes = lib.mae_linear_wgrad_pair_scratch_bytes(M, s0[0], s0[1], s1[0], s1[1])
    assert nbytes >= max(lib.mae_linear_wgrad_scratch_bytes(M, *s0), lib.mae_linear_wgrad_scratch_bytes(M, *s1))
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    (y0, a0, N0, K0, w0, b0), (y1, a1, N1, K1, w1, b1) = args
    for rep in range(2):
        check(lib.mae_linear_wgrad_pair(_ptr(y0), _ptr(a0), N0, K0, _ptr(w0), _ptr(b0), _ptr(y1), _ptr(a1), N1, K1, _ptr(w1), _ptr(b1),
                                        M, dt, _ptr(scratch), stream(dev)))
    for dW, db, ref, refb in outs:
        assert rel_err(dW, ref) < 2e-5 and rel_err(db, refb) < 2e-5


@pytest.mark.parametrize("M,N,K", [(8200, 1536, 384), (6000, 384, 1536), (4289, 576, 192), (20000, 1152, 384), (4097, 192, 192), (5000, 512, 1024),
                                   (4100, 328, 200), (4931, 768, 768), (577, 384, 192)])
def test_wgrad_buffer_dma_kernel_is_bit_identical_to_the_pointer_dma_kernel(dev, M, N, K, monkeypatch):
    """Round 3's gemm_tn4_kernel (range-checked buffer DMA, no clamp / zero-fill pass for a ragged last step, pieces issued between
    MFMA thirds or in a burst) sums the same products in the same order as gemm_tn3_kernel: every M-split's ragged tail, a last
    tile column that sticks out of the matrix, and the bias column agree bit for bit."""
    g = G(M * 3 + N + K)
    dY = dv(torch.randn(M, N, generator=g).to(torch.bfloat16)); A = dv(torch.randn(M, K, generator=g).to(torch.bfloat16))
    scratch = torch.zeros(lib.mae_linear_wgrad_scratch_bytes(M, N, K), dtype=torch.uint8, device=dev)
    res = {}
    for v in ("v3r", "v4", "v4b"):
        monkeypatch.setenv("MAE_WGRAD", v)
        dW = torch.full((N, K), float("nan"), device=dev); db = torch.full((N,), float("nan"), device=dev)
        check(lib.mae_linear_wgrad(_ptr(dY), _ptr(A), M, N, K, BF16, _ptr(dW), _ptr(db), _ptr(scratch), stream(dev)))
        torch.cuda.synchronize()
        res[v] = (dW, db)
    assert not torch.isnan(res["v3r"][0]).any()
    for v in ("v4", "v4b"):
        assert torch.equal(res[v][0], res["v3r"][0]) and torch.equal(res[v][1], res["v3r"][1]), v


def test_wgrad_buffer_dma_32bit_offsets_at_the_top_of_their_range(dev, monkeypatch):
    """gemm_tn4_kernel addresses dY and X with 32-bit byte offsets (descriptor per M-split); a 4.27 GB dY, 0.6 % below the limit the launcher
    admits, gives the bits of the pointer-DMA kernel."""
    M, N, K = 1390000, 1536, 192
    g = torch.Generator(device=dev).manual_seed(3)
    dY = (torch.rand(M, N, device=dev, generator=g) * 2 - 1).to(torch.bfloat16); A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    scratch = torch.zeros(lib.mae_linear_wgrad_scratch_bytes(M, N, K), dtype=torch.uint8, device=dev)
    res = {}
    for v in ("v3r", "v4"):
        monkeypatch.setenv("MAE_WGRAD", v)
        dW = torch.full((N, K), float("nan"), device=dev); db = torch.full((N,), float("nan"), device=dev)
        check(lib.mae_linear_wgrad(_ptr(dY), _ptr(A), M, N, K, BF16, _ptr(dW), _ptr(db), _ptr(scratch), stream(dev)))
        torch.cuda.synchronize()
        res[v] = (dW, db)
    assert torch.isfinite(res["v4"][0]).all() and torch.equal(res["v4"][0], res["v3r"][0]) and torch.equal(res["v4"][1], res["v3r"][1])


@pytest.mark.parametrize("M,s0,s1", [(9000, (384, 1536), (1536, 384)), (8300, (384, 384), (1152, 384)), (8197, (192, 192), (576, 192)), (9001, (512, 1024), (1536, 512))])
def test_wgrad_pair_buffer_dma_kernel_is_bit_identical(dev, M, s0, s1, monkeypatch):
    g = G(M + s0[0])
    ops = []
    for N, K in (s0, s1):
        ops.append((dv(torch.randn(M, N, generator=g).to(torch.bfloat16)), dv(torch.randn(M, K, generator=g).to(torch.bfloat16)), N, K))
    scratch = torch.zeros(lib.mae_linear_wgrad_pair_scratch_bytes(M, *s0, *s1), dtype=torch.uint8, device=dev)
    res = {}
    for mode in ("3", "4", "4b"):
        monkeypatch.setenv("MAE_WGRAD_PAIR", mode)
        outs = [(torch.full((N, K), float("nan"), device=dev), torch.full((N,), float("nan"), device=dev)) for _, _, N, K in ops]
        (y0, a0, N0, K0), (y1, a1, N1, K1) = ops
        check(lib.mae_linear_wgrad_pair(_ptr(y0), _ptr(a0), N0, K0, _ptr(outs[0][0]), _ptr(outs[0][1]), _ptr(y1), _ptr(a1), N1, K1, _ptr(outs[1][0]), _ptr(outs[1][1]),
                                        M, BF16, _ptr(scratch), stream(dev)))
        torch.cuda.synchronize()
        res[mode] = outs
    assert not any(torch.isnan(t).any() for pr in res["3"] for t in pr)
    for mode in ("4", "4b"):
        for (w, b), (w3, b3) in zip(res[mode], res["3"]):
            assert torch.equal(w, w3) and torch.equal(b, b3), mode


# ----------------------------------------------------------------------------------------------- attention
def _attn_ref(qkv, B, T, H, hd):
    q, k, v = qkv.reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    s = (q @ k.transpose(-2, -1)) * hd ** -0.5
    p = s.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(B, T, H * hd), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,T,H,hd", [(3, 36, 6, 24), (2, 145, 6, 32), (5, 36, 6, 64), (2, 145, 8, 64), (1, 17, 2, 24), (2, 72, 6, 64), (1, 300, 2, 32), (4, 1, 2, 32),
                                      (2, 257, 4, 64), (1, 304, 2, 64), (1, 480, 2, 32),   # backward as two launches (four images exceed the LDS)
                                      (1, 500, 2, 64), (1, 700, 2, 32), (1, 600, 1, 64)])  # forward with Q from global memory (three images exceed it); beyond that the block-streamed fallback
@pytest.mark.parametrize("dt", [F32, BF16])
def test_attention_fwd_bwd(dev, B, T, H, hd, dt):
    g = G(T * 3 + hd)
    qkv = (torch.randn(B, T, 3 * H * hd, generator=g)).to(TDT[dt])
    do = torch.randn(B, T, H * hd, generator=g).to(TDT[dt])
    qr = qkv.double().requires_grad_(True)
    oref, lref = _attn_ref(qr, B, T, H, hd)
    oref.backward(do.double())
    out = torch.full((B, T, H * hd), float("nan"), dtype=TDT[dt], device=dev)
    lse = torch.empty(B, H, T, device=dev)
    check(lib.mae_attention_fwd(_ptr(dv(qkv)), B, T, H, hd, dt, _ptr(out), _ptr(lse), stream(dev)))
    tol = 2e-5 if dt == F32 else 8e-3
    assert rel_err(out.float(), oref) < tol
    assert max_err(lse, lref) < (1e-4 if dt == F32 else 2e-2)
    dqkv = torch.full((B, T, 3 * H * hd), float("nan"), dtype=TDT[dt], device=dev)
    check(lib.mae_attention_bwd(_ptr(dv(qkv)), _ptr(out), _ptr(dv(do)), _ptr(lse), B, T, H, hd, dt, _ptr(dqkv), stream(dev)))
    assert rel_err(dqkv.float(), qr.grad) < (5e-5 if dt == F32 else 2e-2)


# ---------------------------------------------------------------------------------- NT GEMM: round 3's K-loop against round 2's
@pytest.mark.parametrize("shape", [(70000, 1536, 384), (4099, 384, 192), (33000, 1152, 384), (9000, 384, 1536), (20000, 512, 256), (300, 192, 192),
                                   (9000, 512, 1024), (72000, 384, 384), (1, 192, 192), (257, 576, 192)])
def test_nt3_kernel_is_bit_identical_to_nt2(dev, shape):
    """gemm_nt3_kernel (split LDS rings, pipelined K-step, scalar-offset buffer DMA, phantom pieces past the last tile) keeps
    round 2's tiles, fragment layout, K order and epilogue arithmetic: every output bit must equal gemm_nt2_kernel's
    (MAE_GEMM_NT=v2 pins the old kernel), for each epilogue the engine's bf16 path uses, both output types, full and ragged
    tiles, one and many tiles per workgroup, nk = 3 .. 24."""
    import os
    M, N, K = shape
    g = torch.Generator(device=dev).manual_seed(5)
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    bias = torch.rand(N, device=dev, generator=g)
    MODE = {"none": 0, "gelu_grad": 4, "mul": 5, "gelu_act": 6}
    try:
        for epi, odt, with_bias in [("none", BF16, True), ("none", BF16, False), ("none", F32, True), ("gelu_grad", BF16, True), ("mul", BF16, False), ("gelu_act", BF16, True)]:
            aux = (torch.rand(M, N, device=dev, generator=g) * 4 - 2).to(torch.bfloat16) if epi == "mul" else None
            outs = {}
            for var in ("v2", "v3"):
                os.environ["MAE_GEMM_NT"] = var
                o = torch.full((M, N), float("nan"), dtype=TDT[odt], device=dev)
                o2 = torch.full((M, N), float("nan"), dtype=TDT[odt], device=dev)
                for _rep in range(2):
                    check(lib.mae_linear_fwd(_ptr(A), _ptr(W), _ptr(bias) if with_bias else None, M, N, K, BF16, MODE[epi], odt, _ptr(o),
                                             _ptr(o2) if epi == "gelu_grad" else None, _ptr(aux) if aux is not None else None, stream(dev)))
                torch.cuda.synchronize()
                outs[var] = (o, o2)
            assert torch.isfinite(outs["v3"][0]).all(), (epi, odt)
            assert torch.equal(outs["v2"][0], outs["v3"][0]), (epi, odt, with_bias)
            if epi == "gelu_grad":
                assert torch.equal(outs["v2"][1], outs["v3"][1]), (epi, odt)
    finally:
        os.environ.pop("MAE_GEMM_NT", None)


@pytest.mark.parametrize("M,N,K,epi,odt", [(1390000, 1536, 192, "none", BF16), (1390000, 768, 192, "none", F32), (1390001, 1536, 192, "mul", BF16)])
def test_nt3_32bit_offsets_at_the_top_of_their_range(dev, M, N, K, epi, odt):
    """The nt3 kernel addresses operands, outputs and side inputs with 32-bit byte offsets inside buffer descriptors (rows past M are dropped by
    the range check).  Shapes whose last tile ends a fraction of a percent below 4 GiB: the result equals the round-2 kernel's (64-bit
    pointers) bit for bit, and the guard rows allocated behind the output stay untouched (a wrapped or unchecked offset would land there
    or at the start of the tensor)."""
    import os
    g = torch.Generator(device=dev).manual_seed(9)
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    aux = (torch.rand(M, N, device=dev, generator=g) * 4 - 2).to(torch.bfloat16) if epi == "mul" else None
    guard = 600                                      # rows behind the tensor (a 256-row tile may start up to 255 rows before the end)
    outs = {}
    try:
        for var in ("v2", "v3"):
            os.environ["MAE_GEMM_NT"] = var
            buf = torch.full((M + guard, N), 3.0, dtype=TDT[odt], device=dev)
            check(lib.mae_linear_fwd(_ptr(A), _ptr(W), None, M, N, K, BF16, {"none": 0, "mul": 5}[epi], odt, _ptr(buf), None, _ptr(aux) if aux is not None else None, stream(dev)))
            torch.cuda.synchronize()
            assert bool((buf[M:] == 3.0).all()), var   # nothing behind row M - 1
            outs[var] = buf[:M]
        assert torch.isfinite(outs["v3"]).all() and torch.equal(outs["v2"], outs["v3"])
    finally:
        os.environ.pop("MAE_GEMM_NT", None)


# ---------------------------------------------------------------------------------- hand-counted vmcnt waits
def test_counted_vmcnt_waits_equal_the_all_drained_build(dev, tmp_path):
    """The persistent NT GEMM and the wgrad ring wait with hand-counted `s_waitcnt vmcnt(N)` (next stage's DMAs + the previous
    tile's epilogue stores stay in flight).  An alternate build with every wait drained to zero (-DMAE_DBG_VMCNT0) must
    produce the same bits for every epilogue mode, both output types, partial tiles and nk <= 2: a miscounted wait would
    read a stage before its DMA has landed."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    dbg = root / "ssrl_vit_mae_jepa_amd" / "lib_dbg_vmcnt0" / "libmae_hip.so"
    subprocess.run(["bash", str(root / "tools" / "build_dbg_lib.sh"), "vmcnt0"], check=True)  # incremental: a no-op when up to date
    script = r"""
import sys, torch
sys.path.insert(0, %r)
from tests.util import BF16, F32, TDT, check, lib, stream, _ptr
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(11)
MODE = {"none": 0, "gelu": 1, "resid": 2, "dgelu": 3, "gelu_grad": 4, "mul": 5, "gelu_act": 6}
out = {}
# (M, N, K): full tiles, a ragged last tile, several tiles per CU, nk = 3 / 6 / 24, N multiples of 192 and of 128
for M, N, K in [(70000, 1536, 384), (4099, 384, 192), (33000, 1152, 384), (9000, 384, 1536), (20000, 512, 256), (300, 192, 192), (9000, 512, 1024)]:
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    bias = torch.rand(N, device=dev, generator=g)
    for epi in MODE:
        for odt in (BF16, F32):
            if epi == "resid" and odt == BF16:
                continue
            aux = None
            if epi == "resid":
                aux = torch.rand(M, N, device=dev, generator=g)
            elif epi in ("dgelu", "mul"):
                aux = (torch.rand(M, N, device=dev, generator=g) * 4 - 2).to(TDT[odt])
            o = torch.zeros(M, N, dtype=TDT[odt], device=dev); o2 = torch.zeros_like(o)
            for rep in range(3):  # back-to-back launches: stages and stores of the previous launch still in flight
                check(lib.mae_linear_fwd(_ptr(A), _ptr(W), _ptr(bias), M, N, K, BF16, MODE[epi], odt, _ptr(o),
                                         _ptr(o2) if epi in ("gelu", "gelu_grad") else None, _ptr(aux) if aux is not None else None, stream(dev)))
            out[f"nt/{M}x{N}x{K}/{epi}/{odt}"] = (o.float().cpu(), o2.float().cpu())
    dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    dY = (torch.rand(M, N, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    scratch = torch.empty(max(1, lib.mae_linear_wgrad_scratch_bytes(M, N, K)), dtype=torch.uint8, device=dev)
    for rep in range(2):
        check(lib.mae_linear_wgrad(_ptr(dY), _ptr(A), M, N, K, BF16, _ptr(dW), _ptr(db), _ptr(scratch), stream(dev)))
    out[f"tn/{M}x{N}x{K}"] = (dW.cpu(), db.cpu())
# two weight gradients in one launch (the engine's pairing of a block's fc2 + fc1)
M = 20000
y0 = (torch.rand(M, 384, device=dev, generator=g) * 2 - 1).to(torch.bfloat16); a0 = (torch.rand(M, 1536, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
y1 = (torch.rand(M, 1536, device=dev, generator=g) * 2 - 1).to(torch.bfloat16); a1 = (torch.rand(M, 384, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
w0 = torch.empty(384, 1536, device=dev); b0 = torch.empty(384, device=dev); w1 = torch.empty(1536, 384, device=dev); b1 = torch.empty(1536, device=dev)
scratch = torch.empty(lib.mae_linear_wgrad_pair_scratch_bytes(M, 384, 1536, 1536, 384), dtype=torch.uint8, device=dev)
for rep in range(2):
    check(lib.mae_linear_wgrad_pair(_ptr(y0), _ptr(a0), 384, 1536, _ptr(w0), _ptr(b0), _ptr(y1), _ptr(a1), 1536, 384, _ptr(w1), _ptr(b1), M, BF16, _ptr(scratch), stream(dev)))
out["tn_pair/fc2"] = (w0.cpu(), b0.cpu()); out["tn_pair/fc1"] = (w1.cpu(), b1.cpu())
torch.cuda.synchronize()
torch.save(out, sys.argv[1])
""" % str(root)
    res = {}
    for name, libpath in (("product", None), ("vmcnt0", dbg)):
        env = dict(os.environ)
        env.pop("MAE_HIP_LIB", None)
        if libpath is not None:
            env["MAE_HIP_LIB"] = str(libpath)
        f = tmp_path / f"{name}.pt"
        r = subprocess.run([sys.executable, "-c", script, str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[name] = torch.load(f, weights_only=True)
    assert set(res["product"]) == set(res["vmcnt0"]) and len(res["product"]) > 60
    for k, (a, b) in res["product"].items():
        a2, b2 = res["vmcnt0"][k]
        assert torch.equal(a, a2) and torch.equal(b, b2), k
        assert torch.isfinite(a).all() and float(a.abs().sum()) > 0, k
