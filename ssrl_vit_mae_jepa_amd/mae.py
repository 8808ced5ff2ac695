"""MaskedAutoencoder: host-side mirror of the reference's ``src/models/mae.py`` (lines 12-94) over libmae_hip.so.

Same constructor dicts, attributes (``mask_ratio, image_size, patch_size, in_chans, sequence_length, encoder,
decoder``), methods (``forward``, ``forward_encoder``, ``forward_decoder``) and ``state_dict`` key names, so
``scripts.training.pretrain_mae`` / ``src.training.mae.MAEPretrainModule`` can use it in place of the lightly/timm
model.  PyTorch supplies device memory, streams and the autograd hook-up only; every arithmetic step of the path runs
in the HIP library.  There is no CPU path: calling ``forward`` on CPU tensors raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _lib
from ._lib import MaeConfig, check, lib


def _stream(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_ptr = _lib.ptr  # keeps the tensor referenced until the call it is an argument of has returned


class Engine:
    """Thin owner of one ``mae_engine_t`` handle plus its parameter table."""

    def __init__(self, cfg: Dict[str, int], precision: str):
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        self.precision = precision
        self.act = _lib.MAE_BF16 if precision == "bf16" else _lib.MAE_F32
        c = MaeConfig(act_dtype=self.act, mlp_ratio=4, **cfg)
        h = C.c_void_p()
        try:
            check(lib.mae_engine_create(C.byref(c), C.byref(h)))
        except _lib.MaeHipError as exc:  # constructor errors of the reference are ValueError/AssertionError
            raise ValueError(str(exc)) from None
        self.handle = h
        self.arena_elems = lib.mae_engine_arena_elems(h)
        self.trainable_elems = lib.mae_engine_trainable_elems(h)
        self.wcache_bytes = lib.mae_engine_wcache_bytes(h)
        self.table: List[Tuple[str, int, int, Tuple[int, ...], int]] = []
        for i in range(lib.mae_engine_num_params(h)):
            name, off, numel, ndim, flags = C.c_char_p(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
            shape = (C.c_int64 * 4)()
            check(lib.mae_engine_param_info(h, i, C.byref(name), C.byref(off), C.byref(numel), C.byref(ndim), shape, C.byref(flags)))
            self.table.append((name.value.decode(), off.value, numel.value, tuple(shape[: ndim.value]), flags.value))

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            lib.mae_engine_destroy(h)

    def workspace_bytes(self, batch: int, num_keep: int) -> int:
        n = lib.mae_engine_workspace_bytes(self.handle, batch, num_keep)
        if n < 0:
            raise ValueError(f"bad batch/num_keep ({batch}, {num_keep})")
        return n

    # timers -------------------------------------------------------------------------------------
    def timers_enable(self, on: bool) -> None:
        check(lib.mae_engine_timers_enable(self.handle, int(on)))

    def timers_reset(self) -> None:
        check(lib.mae_engine_timers_reset(self.handle))

    def timers_read(self) -> Dict[str, Dict[str, float]]:
        out = {}
        for k in range(lib.mae_engine_timer_count(self.handle)):
            ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            check(lib.mae_engine_timer_read(self.handle, k, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
            out[lib.mae_engine_timer_name(self.handle, k).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
        return out


class _Node(nn.Module):
    """Parameter holder: gives the flat arena the reference's module/attribute names."""


class _ViT(_Node):
    """Stands where timm's VisionTransformer stands (``mae.encoder.vit``): ``blocks`` (indexable), ``norm``,
    ``embed_dim``, ``forward_features`` -- what the reference's classifier touches (src/models/classifier.py:47-57,
    src/training/classifier.py:144-165)."""
    embed_dim: int = 0

    def forward_features(self, images: torch.Tensor) -> torch.Tensor:
        """timm VisionTransformer.forward_features: every token, no masking (scripts/training/train_mae.py:143).
        Differentiable w.r.t. the encoder parameters (one autograd node over the engine's encoder backward)."""
        return self._owner().forward_encoder(images, idx_keep=None)

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        return self.forward_features(images)


class _Encoder(_Node):
    """lightly MaskedVisionTransformerTIMM surface used at src/models/mae.py:55."""

    def encode(self, images: torch.Tensor, idx_keep: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self._owner().forward_encoder(images, idx_keep=idx_keep)


class _Decoder(_Node):
    """lightly MAEDecoderTIMM surface used at src/models/mae.py:59-73: embed / decode / predict.  The three are
    engine-backed inference calls; the differentiable route through the decoder is ``MaskedAutoencoder.forward_decoder``
    (or ``forward``), so calling them where autograd would have to record raises instead of detaching silently."""

    def _guard(self, what: str, x: torch.Tensor) -> "MaskedAutoencoder":
        m = self._owner()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in m._dec_params)):
            raise RuntimeError(f"decoder.{what}() is an inference call on the MI355X engine and records no autograd graph; "
                               "use MaskedAutoencoder.forward_decoder()/forward() for gradients, or wrap the call in torch.no_grad()")
        return m

    def embed(self, x: torch.Tensor) -> torch.Tensor:
        """decoder_embed Linear(D, Dd) (src/models/mae.py:59): (B, n, D) -> (B, n, Dd)."""
        m = self._guard("embed", x)
        return m._linear("decoder.decoder_embed", x, round_to_act=True)

    def decode(self, x: torch.Tensor) -> torch.Tensor:
        """+ decoder_pos_embed -> decoder blocks -> decoder_norm (src/models/mae.py:71): (B, L, Dd) -> (B, L, Dd)."""
        m = self._guard("decode", x)
        dev = m._require_cuda()
        B, L, Dd = x.shape
        if L != m.sequence_length or Dd != m._dims["decoder_embed_dim"]:
            raise ValueError(f"decode expects (B, {m.sequence_length}, {m._dims['decoder_embed_dim']}), got {tuple(x.shape)}")
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        ws = m._ws(B, 1)
        m._gen_enc += 1; m._gen_dec += 1  # the block buffers of the workspace are overwritten
        out = torch.empty_like(x)
        check(lib.mae_engine_decoder_decode(m._engine.handle, _ptr(m._arena), _ptr(m._weights()), _ptr(x), B, _ptr(ws), ws.numel(),
                                           _ptr(out), _stream(dev)))
        return out

    def predict(self, x: torch.Tensor) -> torch.Tensor:
        """decoder_pred Linear(Dd, p*p*C) (src/models/mae.py:73): (B, n, Dd) -> (B, n, P)."""
        m = self._guard("predict", x)
        return m._linear("decoder.decoder_pred", x, round_to_act=False)


class _MAEFunction(torch.autograd.Function):
    """One autograd node for the whole MaskedAutoencoder.forward: backward runs the engine's fused backward."""

    @staticmethod
    def forward(ctx, model: "MaskedAutoencoder", images, idx_keep, idx_mask, *params):
        x_pred, target = model._run_forward(images, idx_keep, idx_mask)
        ctx.model = model
        ctx.generation = (model._gen_enc, model._gen_dec)
        ctx.dims = (images.shape[0], idx_keep.shape[1], idx_mask.shape[1])
        ctx.mark_non_differentiable(target)
        return x_pred, target

    @staticmethod
    def backward(ctx, g_pred, _g_target):
        model = ctx.model
        if ctx.generation != (model._gen_enc, model._gen_dec):
            raise RuntimeError(_STALE)
        grads = model._run_backward(g_pred.contiguous().float(), *ctx.dims)
        return (None, None, None, None, *grads)


_STALE = ("MaskedAutoencoder: another forward overwrote the saved activations before backward; "
          "run eval/validation forwards under torch.no_grad() or after backward")


class _EncoderFunction(torch.autograd.Function):
    """forward_encoder / forward_features as an autograd node: backward = mae_engine_backward_encoder."""

    @staticmethod
    def forward(ctx, model: "MaskedAutoencoder", images, idx_keep, *params):
        out = model._run_encoder(images, idx_keep)
        ctx.model, ctx.generation, ctx.dims = model, model._gen_enc, (images.shape[0], idx_keep.shape[1])
        return out

    @staticmethod
    def backward(ctx, g):
        model = ctx.model
        if ctx.generation != model._gen_enc:
            raise RuntimeError(_STALE)
        dev = model._require_cuda()
        B, k = ctx.dims
        ws = model._ws(B, k, keep=True)
        buf = torch.zeros(model._engine.trainable_elems, dtype=torch.float32, device=dev)
        check(lib.mae_engine_backward_encoder(model._engine.handle, _ptr(model._arena), _ptr(model._weights()), _ptr(g.contiguous().float()),
                                             B, k, _ptr(ws), ws.numel(), _ptr(buf), _stream(dev)))
        return (None, None, None, *model._views(buf, model._enc_table))


class _DecoderFunction(torch.autograd.Function):
    """forward_decoder as an autograd node: backward = mae_engine_backward_decoder (decoder gradients + d x_encoded)."""

    @staticmethod
    def forward(ctx, model: "MaskedAutoencoder", x_encoded, idx_keep, idx_mask, *params):
        out = model._run_decoder(x_encoded, idx_keep, idx_mask)
        ctx.model, ctx.generation = model, model._gen_dec
        ctx.dims = (x_encoded.shape[0], idx_keep.shape[1], idx_mask.shape[1])
        return out

    @staticmethod
    def backward(ctx, g):
        model = ctx.model
        if ctx.generation != model._gen_dec:
            raise RuntimeError(_STALE)
        dev = model._require_cuda()
        B, k, m = ctx.dims
        ws = model._ws(B, k, keep=True)
        buf = torch.zeros(model._engine.trainable_elems, dtype=torch.float32, device=dev)
        dx = torch.empty(B, k, model._dims["embed_dim"], dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        check(lib.mae_engine_backward_decoder(model._engine.handle, _ptr(model._arena), _ptr(model._weights()), _ptr(g.contiguous().float()),
                                             B, k, m, _ptr(ws), ws.numel(), _ptr(buf), _ptr(dx), _stream(dev)))
        return (None, dx, None, None, *model._views(buf, model._dec_table))


class MaskedAutoencoder(nn.Module):
    """Masked Autoencoder (MAE) with ViT backbone -- reference API, MI355X engine."""

    def __init__(self, general_cfg: Dict[str, Any], encoder_cfg: Dict[str, Any], decoder_cfg: Dict[str, Any]):
        super().__init__()
        # defaults exactly as the reference reads them (src/models/mae.py:23-26, 32-34, 49-51)
        self.mask_ratio = general_cfg.get("mask_ratio", 0.75)
        self.image_size = general_cfg.get("image_size", 96)
        self.patch_size = general_cfg.get("patch_size", 6)
        self.in_chans = general_cfg.get("in_chans", 3)
        precision = general_cfg.get("engine_precision", os.environ.get("MAE_HIP_PRECISION", "bf16"))
        cfg = dict(
            image_size=int(self.image_size), patch_size=int(self.patch_size), in_chans=int(self.in_chans),
            embed_dim=int(encoder_cfg.get("embed_dim", 384)), depth=int(encoder_cfg.get("depth", 12)),
            num_heads=int(encoder_cfg.get("num_heads", 6)),
            decoder_embed_dim=int(decoder_cfg.get("decoder_embed_dim", 512)),
            decoder_depth=int(decoder_cfg.get("decoder_depth", 4)),
            decoder_num_heads=int(decoder_cfg.get("decoder_num_heads", 6)),
            pred_dim=int(general_cfg.get("pred_dim", 0)),  # 0 = pixels (MAE); the I-JEPA net sets the encoder width
        )
        self._dims = cfg
        self._engine = Engine(cfg, precision)
        self.sequence_length = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
        self.patch_dim = cfg["pred_dim"] or cfg["patch_size"] ** 2 * cfg["in_chans"]  # width of the prediction head

        self.encoder = _Encoder()
        self.decoder = _Decoder()
        self.encoder.vit = _ViT()
        self.encoder.vit.embed_dim = cfg["embed_dim"]
        owner = [self]  # no module cycle in nn.Module registration
        for n in (self.encoder, self.encoder.vit, self.decoder):
            object.__setattr__(n, "_owner", lambda o=owner: o[0])

        self._arena = torch.zeros(self._engine.arena_elems, dtype=torch.float32)
        self._trainable: List[nn.Parameter] = []
        self._slots: List[Tuple[nn.Parameter, int, int, Tuple[int, ...]]] = []
        for name, off, numel, shape, flags in self._engine.table:
            p = nn.Parameter(self._arena[off:off + numel].view(shape), requires_grad=not (flags & _lib.PARAM_FROZEN))
            self._register(name, p)
            self._slots.append((p, off, numel, shape))
            if flags & _lib.PARAM_TRAINABLE:
                self._trainable.append(p)
        self._init_weights()
        # the two halves of the gradient arena (encoder tensors first): parameter lists of the per-half autograd nodes
        split = lib.mae_engine_encoder_grad_elems(self._engine.handle)
        rows = [(row, p) for row, (p, *_r) in zip(self._engine.table, self._slots) if row[4] & _lib.PARAM_TRAINABLE]
        self._enc_table = [row for row, _p in rows if row[1] < split]
        self._dec_table = [row for row, _p in rows if row[1] >= split]
        self._enc_params = [p for row, p in rows if row[1] < split]
        self._dec_params = [p for row, p in rows if row[1] >= split]
        self._offsets = {row[0]: row for row in self._engine.table}

        self._grad_arena: Optional[torch.Tensor] = None
        self._wcache: Optional[torch.Tensor] = None
        self._wcache_version = -1
        self._workspace: Optional[torch.Tensor] = None
        # saved-activation bookkeeping: a forward bumps the generation of what it overwrites; a backward whose node saw
        # an older generation refuses to run on someone else's activations
        self._gen_enc = 0
        self._gen_dec = 0
        self._plan: Optional[Tuple[int, int]] = None
        self._scratch: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ construction helpers
    def _register(self, dotted: str, p: nn.Parameter) -> None:
        node: nn.Module = self
        parts = dotted.split(".")
        for part in parts[:-1]:
            child = node._modules.get(part)
            if child is None:
                # timm keeps the blocks in an nn.Sequential: len(), indexing and slicing are part of the surface
                child = nn.ModuleList() if part in ("blocks", "decoder_blocks") else _Node()
                node.add_module(part, child)
            node = child
        node.register_parameter(parts[-1], p)

    @torch.no_grad()
    def _init_weights(self, seed: Optional[int] = None) -> None:
        """lightly/timm recipe: xavier-uniform Linear (patch projection on its 2-D view), zero bias, LayerNorm 1/0,
        tokens N(0, .02), frozen 2-D sin-cos position tables (lightly MaskedVisionTransformerTIMM / MAEDecoderTIMM)."""
        g = None
        if seed is not None:
            g = torch.Generator().manual_seed(seed)
        for (name, _off, _n, shape, flags), (p, *_r) in zip(self._engine.table, self._slots):
            if name.endswith("pos_embed"):
                p.copy_(_sincos_2d(shape[-1], self.image_size // self.patch_size))
            elif name.endswith("_token"):
                p.copy_(torch.randn(shape, generator=g) * 0.02)
            elif flags & _lib.PARAM_MATRIX:
                fan_out, fan_in = shape[0], int(torch.tensor(shape[1:]).prod())
                bound = (6.0 / (fan_in + fan_out)) ** 0.5
                p.copy_((torch.rand(shape, generator=g) * 2 - 1) * bound)
            elif name.endswith("weight"):
                p.fill_(1.0)  # LayerNorm gain
            else:
                p.zero_()

    def _apply(self, fn, recurse=True):
        super()._apply(fn)
        self._reflatten()
        return self

    @torch.no_grad()
    def _reflatten(self) -> None:
        """After .to()/.cuda(): gather the parameters back into one flat fp32 arena on their new device."""
        first = self._slots[0][0]
        device = first.device
        if first.dtype == torch.float32 and self._arena.device == device and first.data_ptr() == self._arena.data_ptr() + 4 * self._slots[0][1]:
            return
        arena = torch.zeros(self._engine.arena_elems, dtype=torch.float32, device=device)
        for p, off, numel, shape in self._slots:
            view = arena[off:off + numel].view(shape)
            view.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = view
            p.grad = None
        self._arena = arena
        self._grad_arena = self._wcache = self._workspace = self._scratch = None
        self._wcache_version = -1

    # ------------------------------------------------------------------ device-side state
    @property
    def engine(self) -> Engine:
        return self._engine

    @property
    def flat_params(self) -> torch.Tensor:
        return self._arena

    GRAD_TAIL = 64 + 64 * 64  # floats behind the gradients: slot 0 carries the step's loss through the data-parallel all-reduce; the rest pads the
    #                           arena to world equal shards for the sharded optimizer (MAE_DP_SHARDED_OPT=1: up to 64 ranks of 64-float granules)

    @property
    def grad_buffer(self) -> torch.Tensor:
        """Gradient arena plus GRAD_TAIL floats (one allocation, so a bucket can cover gradients and the loss slot)."""
        if self._grad_arena is None or self._grad_arena.device != self._arena.device:
            self._grad_arena = torch.zeros(self._engine.trainable_elems + self.GRAD_TAIL, dtype=torch.float32, device=self._arena.device)
        return self._grad_arena

    @property
    def flat_grads(self) -> torch.Tensor:
        return self.grad_buffer[:self._engine.trainable_elems]

    def _require_cuda(self) -> torch.device:
        dev = self._arena.device
        if dev.type != "cuda":
            raise RuntimeError("MaskedAutoencoder runs on MI355X only: call .cuda() first (there is no CPU fallback)")
        return dev

    def _params_version(self) -> int:
        return sum(p._version for p, *_ in self._slots)

    def mark_weights_fresh(self) -> None:
        self._wcache_version = self._params_version()

    def _weights(self) -> Optional[torch.Tensor]:
        """bf16 / transposed operand copies of the GEMM weights, refreshed when a parameter changed."""
        dev = self._require_cuda()
        if self._wcache is None:
            self._wcache = torch.zeros(self._engine.wcache_bytes, dtype=torch.uint8, device=dev)
            self._wcache_version = -1
        if self._engine.act == _lib.MAE_BF16:
            v = self._params_version()
            if v != self._wcache_version:
                check(lib.mae_engine_refresh_weights(self._engine.handle, _ptr(self._arena), _ptr(self._wcache), _stream(dev)))
                self._wcache_version = v
        return self._wcache

    def _ws(self, batch: int, num_keep: int, keep: bool = False) -> torch.Tensor:
        """The workspace for (batch, num_keep).  Its layout is a function of that pair, so a forward with another pair
        (or a reallocation) invalidates every saved activation; ``keep`` = a backward asking for its own forward's plan."""
        need = self._engine.workspace_bytes(batch, num_keep)
        realloc = self._workspace is None or self._workspace.numel() < need or self._workspace.device != self._arena.device
        if keep and (realloc or self._plan != (batch, num_keep)):
            raise RuntimeError(_STALE)
        if realloc:
            self._workspace = None  # release before growing
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self._arena.device)
        if realloc or self._plan != (batch, num_keep):
            self._gen_enc += 1; self._gen_dec += 1
            self._plan = (batch, num_keep)
        return self._workspace

    def _views(self, flat: torch.Tensor, table) -> List[torch.Tensor]:
        return [flat[off:off + numel].view(shape) for (_name, off, numel, shape, _flags) in table]

    def _linear(self, prefix: str, x: torch.Tensor, round_to_act: bool) -> torch.Tensor:
        """One Linear of the model through mae_linear_fwd (the engine's GEMM, operands in the engine's precision)."""
        dev = self._require_cuda()
        _n, w_off, w_numel, w_shape, _f = self._offsets[prefix + ".weight"]
        _n, b_off, b_numel, _s, _f = self._offsets[prefix + ".bias"]
        N, K = w_shape[0], w_numel // w_shape[0]
        if x.shape[-1] != K:
            raise ValueError(f"{prefix}: expected last dim {K}, got {tuple(x.shape)}")
        bf = self._engine.act == _lib.MAE_BF16
        a = x.to(device=dev, dtype=torch.bfloat16 if bf else torch.float32).contiguous().view(-1, K)
        w = self._weights()[2 * w_off:2 * (w_off + w_numel)] if bf else self._arena[w_off:w_off + w_numel]
        out_dt = self._engine.act if round_to_act else _lib.MAE_F32
        out = torch.empty(a.shape[0], N, dtype=torch.bfloat16 if (bf and round_to_act) else torch.float32, device=dev)
        check(lib.mae_linear_fwd(_ptr(a), _ptr(w), _ptr(self._arena[b_off:b_off + b_numel]), a.shape[0], N, K, self._engine.act,
                                _lib.EPI_NONE, out_dt, _ptr(out), None, None, _stream(dev)))
        return out.float().view(*x.shape[:-1], N)

    def _scratch_f32(self) -> torch.Tensor:
        if self._scratch is None or self._scratch.device != self._arena.device:
            self._scratch = torch.zeros(8192, dtype=torch.float32, device=self._arena.device)
        return self._scratch

    def num_keep(self, mask_ratio: Optional[float] = None) -> int:
        r = self.mask_ratio if mask_ratio is None else mask_ratio
        return max(1, int(self.sequence_length * (1 - r)))  # lightly random_token_mask

    # ------------------------------------------------------------------ reference API
    def random_token_mask(self, batch_size: int, noise: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """lightly utils.random_token_mask (called at src/models/mae.py:79-83): idx_keep, idx_mask int64."""
        dev = self._require_cuda()
        L, k = self.sequence_length, self.num_keep()
        if noise is None:
            noise = torch.rand(batch_size, L, device=dev)
        noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        if noise.shape != (batch_size, L):
            raise ValueError(f"noise must be ({batch_size}, {L}), got {tuple(noise.shape)}")
        keep = torch.empty(batch_size, k, dtype=torch.int64, device=dev)
        mask = torch.empty(batch_size, L - k, dtype=torch.int64, device=dev)
        check(lib.mae_mask_from_noise(_ptr(noise), batch_size, L, k, _ptr(keep), _ptr(mask), _stream(dev)))
        return keep, mask

    def _check_images(self, images: torch.Tensor) -> torch.Tensor:
        if images.dim() != 4 or images.shape[1] != self.in_chans or images.shape[2] != self.image_size or images.shape[3] != self.image_size:
            raise ValueError(f"Input size {tuple(images.shape)} doesn't match model "
                             f"(B, {self.in_chans}, {self.image_size}, {self.image_size})")  # timm PatchEmbed assert
        if images.dtype == torch.uint8:  # raw pixels: ToTensor + Normalize(.5,.5) happens inside the pixel-reading kernels
            return images.contiguous()
        return images.to(dtype=torch.float32).contiguous()

    @staticmethod
    def _img_dt(images: torch.Tensor) -> int:
        return _lib.MAE_U8 if images.dtype == torch.uint8 else _lib.MAE_F32

    def _check_idx(self, idx: torch.Tensor, batch: int, what: str) -> torch.Tensor:
        if idx.dim() != 2 or idx.shape[0] != batch:
            raise ValueError(f"{what} must be (batch, n), got {tuple(idx.shape)}")
        idx = idx.to(dtype=torch.int64).contiguous()
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= self.sequence_length):
            raise IndexError(f"{what} out of range [0, {self.sequence_length})")
        return idx

    def _run_encoder(self, images: torch.Tensor, idx_keep: torch.Tensor) -> torch.Tensor:
        dev = self._require_cuda()
        B, k = images.shape[0], idx_keep.shape[1]
        ws = self._ws(B, k)
        self._gen_enc += 1; self._gen_dec += 1  # enc_norm, which the decoder's backward reads, is rewritten too
        out = torch.empty(B, k, self._dims["embed_dim"], dtype=torch.float32, device=dev)
        check(lib.mae_engine_forward_encoder(self._engine.handle, _ptr(self._arena), _ptr(self._weights()), _ptr(images), self._img_dt(images),
                                            _ptr(idx_keep), B, k, _ptr(ws), ws.numel(), _ptr(out), _stream(dev)))
        return out

    def forward_encoder(self, images: torch.Tensor, idx_keep: Optional[torch.Tensor] = None) -> torch.Tensor:
        """src/models/mae.py:54-55 -> (B, k, D).  Differentiable w.r.t. the encoder parameters: with grad enabled the
        call is one autograd node whose backward is the engine's encoder backward (the classifier hand-off,
        scripts/training/train_mae.py:143)."""
        dev = self._require_cuda()
        images = self._check_images(images)
        B = images.shape[0]
        if idx_keep is None:
            idx_keep = torch.arange(self.sequence_length, device=dev).repeat(B, 1)
        idx_keep = self._check_idx(idx_keep.to(dev), B, "idx_keep")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._enc_params):
            return _EncoderFunction.apply(self, images, idx_keep, *self._enc_params)
        return self._run_encoder(images, idx_keep)

    def _run_decoder(self, x_encoded: torch.Tensor, idx_keep: torch.Tensor, idx_mask: torch.Tensor) -> torch.Tensor:
        dev = self._require_cuda()
        B, k, m = x_encoded.shape[0], idx_keep.shape[1], idx_mask.shape[1]
        ws = self._ws(B, k)
        self._gen_dec += 1
        out = torch.empty(B, m, self.patch_dim, dtype=torch.float32, device=dev)
        check(lib.mae_engine_forward_decoder(self._engine.handle, _ptr(self._arena), _ptr(self._weights()), _ptr(x_encoded),
                                            _ptr(idx_keep), _ptr(idx_mask), B, k, m, _ptr(ws), ws.numel(), _ptr(out), _stream(dev)))
        return out

    def forward_decoder(self, x_encoded: torch.Tensor, idx_keep: torch.Tensor, idx_mask: torch.Tensor) -> torch.Tensor:
        """src/models/mae.py:57-75 -> (B, m, p*p*C).  Differentiable w.r.t. x_encoded and the decoder parameters."""
        dev = self._require_cuda()
        B = x_encoded.shape[0]
        idx_keep = self._check_idx(idx_keep.to(dev), B, "idx_keep")
        idx_mask = self._check_idx(idx_mask.to(dev), B, "idx_mask")
        k, m = idx_keep.shape[1], idx_mask.shape[1]
        if tuple(x_encoded.shape) != (B, k, self._dims["embed_dim"]):
            raise ValueError(f"x_encoded must be ({B}, {k}, {self._dims['embed_dim']}), got {tuple(x_encoded.shape)}")
        if k + m != self.sequence_length:
            raise ValueError("idx_keep and idx_mask must partition the sequence")
        x_encoded = x_encoded.to(device=dev, dtype=torch.float32).contiguous()
        if torch.is_grad_enabled() and (x_encoded.requires_grad or any(p.requires_grad for p in self._dec_params)):
            return _DecoderFunction.apply(self, x_encoded, idx_keep, idx_mask, *self._dec_params)
        return self._run_decoder(x_encoded, idx_keep, idx_mask)

    def patchify_gather(self, images: torch.Tensor, idx_mask: torch.Tensor) -> torch.Tensor:
        """utils.patchify + get_at_index(clamp(idx_mask - 1, 0)) (src/models/mae.py:90-92)."""
        dev = self._require_cuda()
        B, m = idx_mask.shape
        target = torch.empty(B, m, self.patch_dim, dtype=torch.float32, device=dev)
        images = self._check_images(images)
        check(lib.mae_patchify_gather(_ptr(images), self._img_dt(images), _ptr(idx_mask, torch.int64), B, self.in_chans, self.image_size,
                                     self.patch_size, m, _ptr(target), _stream(dev)))
        return target

    def _run_forward(self, images, idx_keep, idx_mask):
        dev = self._require_cuda()
        B, k, m = images.shape[0], idx_keep.shape[1], idx_mask.shape[1]
        ws = self._ws(B, k)
        self._gen_enc += 1; self._gen_dec += 1
        h, w = self._engine.handle, self._weights()
        x_pred = torch.empty(B, m, self.patch_dim, dtype=torch.float32, device=dev)
        check(lib.mae_engine_forward_encoder(h, _ptr(self._arena), _ptr(w), _ptr(images), self._img_dt(images), _ptr(idx_keep), B, k, _ptr(ws),
                                            ws.numel(), None, _stream(dev)))
        check(lib.mae_engine_forward_decoder(h, _ptr(self._arena), _ptr(w), None, _ptr(idx_keep), _ptr(idx_mask), B, k, m, _ptr(ws),
                                            ws.numel(), _ptr(x_pred), _stream(dev)))
        return x_pred, self.patchify_gather(images, idx_mask)

    def _run_backward(self, d_pred: torch.Tensor, B: int, k: int, m: int) -> List[torch.Tensor]:
        dev = self._require_cuda()
        ws = self._ws(B, k, keep=True)
        g = torch.zeros(self._engine.trainable_elems, dtype=torch.float32, device=dev)
        check(lib.mae_engine_backward(self._engine.handle, _ptr(self._arena), _ptr(self._weights()), _ptr(d_pred), None, B, k, m,
                                     _ptr(ws), ws.numel(), _ptr(g), _stream(dev)))
        out = []
        for (name, off, numel, shape, flags) in self._engine.table:
            if flags & _lib.PARAM_TRAINABLE:
                out.append(g[off:off + numel].view(shape))
        return out

    def forward(self, images: torch.Tensor, noise: Optional[torch.Tensor] = None):
        """src/models/mae.py:77-94: returns (x_pred, target), both (B, num_masked, p*p*C) fp32."""
        self._require_cuda()
        images = self._check_images(images)
        idx_keep, idx_mask = self.random_token_mask(images.shape[0], noise)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._trainable):
            return _MAEFunction.apply(self, images, idx_keep, idx_mask, *self._trainable)
        return self._run_forward(images, idx_keep, idx_mask)

    # ------------------------------------------------------------------ fused step pieces (used by training.py)
    def grad_ready_points(self) -> List[int]:
        """Arena offsets of the backward pass's gradient-ready points, in the order they are reached: reaching point j
        means ``flat_grads[offsets[j]:]`` is final (decoder first, encoder block 0 / patch projection last)."""
        n = lib.mae_engine_grad_ready_points(self._engine.handle, None, 0)
        offs = (C.c_int64 * n)()
        lib.mae_engine_grad_ready_points(self._engine.handle, offs, n)
        return list(offs)

    def loss_and_grads(self, images: torch.Tensor, noise: torch.Tensor, grad_scale: float = 1.0,
                       return_indices: bool = False, ready_events: Optional[List[Optional[torch.cuda.Event]]] = None,
                       loss_out: Optional[torch.Tensor] = None):
        """zero_grad + mask + forward + MSE + backward in one native call (src/training/mae.py:45-50 + loss.backward()).
        Gradients land in ``flat_grads``; returns the device scalar loss (no host sync).
        ``ready_events``: one entry per gradient-ready point (``None`` = skip); each event is recorded on the current
        stream when its point is reached (the data-parallel step starts that range's all-reduce behind it).
        ``loss_out``: a 1-element fp32 device tensor to receive the loss (default: a fresh one)."""
        dev = self._require_cuda()
        images = self._check_images(images)
        B, L, k = images.shape[0], self.sequence_length, self.num_keep()
        if noise.shape != (B, L) or noise.dtype != torch.float32 or not noise.is_contiguous():
            raise ValueError(f"noise must be a contiguous fp32 ({B}, {L}) tensor")
        ws = self._ws(B, k)
        self._gen_enc += 1; self._gen_dec += 1
        loss = torch.empty(1, dtype=torch.float32, device=dev) if loss_out is None else loss_out
        keep = mask = None
        if return_indices:
            keep = torch.empty(B, k, dtype=torch.int64, device=dev)
            mask = torch.empty(B, L - k, dtype=torch.int64, device=dev)
        if ready_events is None:
            check(lib.mae_engine_loss_and_grads(self._engine.handle, _ptr(self._arena), _ptr(self._weights()), _ptr(images), self._img_dt(images),
                                               _ptr(noise), B, k, float(grad_scale), _ptr(ws), ws.numel(), _ptr(self.flat_grads), _ptr(loss, torch.float32),
                                               _ptr(keep), _ptr(mask), _stream(dev)))
        else:
            evs = (C.c_void_p * len(ready_events))(*[C.c_void_p(ev.cuda_event) if ev is not None else C.c_void_p(0) for ev in ready_events])
            check(lib.mae_engine_loss_and_grads_phased(self._engine.handle, _ptr(self._arena), _ptr(self._weights()), _ptr(images),
                                                      self._img_dt(images), _ptr(noise), B, k, float(grad_scale), _ptr(ws), ws.numel(), _ptr(self.flat_grads),
                                                      _ptr(loss, torch.float32), _ptr(keep), _ptr(mask), evs, len(ready_events), _stream(dev)))
        return (loss, keep, mask) if return_indices else loss

    def named_flat_views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """state_dict-named views of a flat trainable-range buffer (grads, exp_avg, ...)."""
        return {name: flat[off:off + numel].view(shape) for (name, off, numel, shape, flags) in self._engine.table
                if flags & _lib.PARAM_TRAINABLE}


def _sincos_2d(embed_dim: int, grid: int) -> torch.Tensor:
    """2-D sin-cos position table with a zero class-token row, laid out as MAE-official / lightly build it
    (first half of the channels from the w coordinate)."""
    if embed_dim % 4:
        raise ValueError("sin-cos position embedding needs embed_dim % 4 == 0")
    gw, gh = torch.meshgrid(torch.arange(grid, dtype=torch.float32), torch.arange(grid, dtype=torch.float32), indexing="xy")

    def one_d(dim: int, pos: torch.Tensor) -> torch.Tensor:
        omega = 1.0 / (10000.0 ** (torch.arange(dim // 2, dtype=torch.float32) / (dim / 2.0)))
        out = pos.reshape(-1)[:, None] * omega[None, :]
        return torch.cat([out.sin(), out.cos()], dim=1)

    emb = torch.cat([one_d(embed_dim // 2, gw), one_d(embed_dim // 2, gh)], dim=1)
    return torch.cat([torch.zeros(1, embed_dim), emb], dim=0).unsqueeze(0)
