"""Categorical RSSM world model with the reference's class / method / state-dict surface
(lib/world_model.py of ProfessorNova/Dreamer), restructured around the MI355X hot path.

What is the same: module and parameter names (``seq.a_emb``, ``seq.linear``, ``seq.norm``,
``seq.gru``, ``enc.conv.N``, ``enc.fc.0``, ``dyn.mlp.N``, ``dyn.head``, ``rew.*``, ``cont.*``,
``dec.fc.0``, ``dec.deconv.N``), constructor arguments, ``init_state`` / ``step`` / ``loss``
signatures and return values, and every numerical quirk (``seq.norm`` exists but is never
applied, lib/world_model.py:35 vs :69-71).

What is different: ``loss`` does not run T whole steps.  Only the recurrence
``(h, z, a, c) -> GRU -> posterior logits -> z`` is sequential; the encoder convolutions are
hoisted over all B*T frames before it and the prior MLP, reward / continue heads, decoder and the
three loss terms are evaluated once over all B*T states after it (same values; the reference's
per-step ``.mean()`` summed over t and divided by T equals the mean over all (b, t)).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .utils import log_unimix, symlog


@dataclass
class WorldModelState:
    h: torch.Tensor  # (B, hidden) or (B, T, hidden)
    z: torch.Tensor  # (B, L, K) one-hot, or (B, T, L, K)


# From this many rows on, the Linear products of the batched (B*T)-row side of the update (forward, input gradient,
# weight gradient) run on the library's fp32 MFMA kernel; below it the backend's routine serves the few rows.
NATIVE_GEMM_MIN_ROWS = int(os.environ.get("DRM_NATIVE_GEMM_MIN_ROWS", "256"))


# Form of the batched Linear products that are large enough for the large-row kernels (output tiles to fill the device, a
# reduction of 256 or more): "split" = fp32 accuracy from an exact three-way bf16 split of both operands on the bf16 matrix
# pipe (1.25-1.35x the fp32 matrix instructions on these shapes, and a little closer to the fp64 result), "bf16" = bf16
# matrix inputs with fp32 accumulation (set by WorldModel.loss for a wide model whose scan runs that way), "f32" = the fp32
# kernel for everything (DRM_NO_SPLIT_PRODUCTS=1).  Read when a layer's forward runs and remembered for its backward.
_PRODUCT_FORM = "f32" if os.environ.get("DRM_NO_SPLIT_PRODUCTS", "0") == "1" else "split"


class _bf16_products:
    """Context manager: the large products take bf16 matrix inputs inside."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global _PRODUCT_FORM
        self.before = _PRODUCT_FORM
        if self.on:
            _PRODUCT_FORM = "bf16"

    def __exit__(self, *exc):
        global _PRODUCT_FORM
        _PRODUCT_FORM = self.before


# the large Linear gradients in the split form read their operands as they lie (the kernel transposes tiles on their way into
# LDS); DRM_NO_GRADIENT_STAGING=1 goes back to transposed copies and the forward kernel
_GRADIENT_STAGING = os.environ.get("DRM_NO_GRADIENT_STAGING", "0") != "1"


def _large_product(form: str, x, w, b=None):
    from . import _native as N
    return N.fused_linear(x, w, b, bf16=form == "bf16", split=form == "split")


def _large_product_ok(rows: int, cols: int, k: int) -> bool:
    """Output of rows x cols from a reduction of length k: enough 64 x 128 tiles for the device, k a multiple of 32."""
    return rows >= 256 and k >= 256 and k % 32 == 0 and (rows // 64) * (cols // 128) >= 256


def _odd_width_few_rows(rows: int, out_features: int) -> bool:
    """A few hundred rows into a width that is no multiple of 64 (the critic's 255 buckets at the 240 imagined states):
    the backend's choice for that shape is a 256 x 240 macro-tile, six workgroups on the whole device, 127 us for
    63 MFLOP -- the library's kernel takes such a product even below ``NATIVE_GEMM_MIN_ROWS``."""
    return rows >= 128 and out_features % 64 != 0


def _mm_forward(x, w, b, form="f32"):
    """x @ w.T + b for a Linear weight w (out, in)."""
    from . import _native as N
    if (form != "f32" and x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1 and w.stride(1) == 1
            and _large_product_ok(x.shape[0], w.shape[0], x.shape[1])):
        return _large_product(form, x, w, b)
    if (x.shape[0] >= NATIVE_GEMM_MIN_ROWS or _odd_width_few_rows(x.shape[0], w.shape[0])) and N.gemm_ok(x, w, False, False):
        return N.gemm(x, w, False, False, bias=b)
    return torch.addmm(b, x, w.t()) if b is not None else x @ w.t()


def _mm_dgrad(dy, w, form="f32"):
    """dy @ w: the input gradient of a Linear with weight w (out, in)."""
    from . import _native as N
    if form != "f32" and dy.is_cuda and dy.dtype == torch.float32 and dy.stride(1) == 1 and _large_product_ok(dy.shape[0], w.shape[1], dy.shape[1]):
        if form in ("split", "bf16") and _GRADIENT_STAGING and N.gemm_split_ok(dy, w, False):
            return N.gemm_split(dy, w, False, bf16=form == "bf16")    # w read as it lies, (out, in) = (K, N)
        return _large_product(form, dy, w.t().contiguous())       # the reduction runs over w's rows: transposed once
    if dy.shape[0] >= NATIVE_GEMM_MIN_ROWS and N.gemm_ok(dy, w, False, True):
        return N.gemm(dy, w, False, True)
    return dy @ w


def _mm_wgrad(dy, x, form="f32"):
    """dy.T @ x: the weight gradient of a Linear, a sum over the rows."""
    from . import _native as N
    if form != "f32" and dy.is_cuda and dy.dtype == torch.float32 and _large_product_ok(dy.shape[1], x.shape[1], dy.shape[0]):
        if form in ("split", "bf16") and _GRADIENT_STAGING and N.gemm_split_ok(dy, x, True):
            return N.gemm_split(dy, x, True, bf16=form == "bf16")     # both operands read as they lie, rows = the reduction
        return _large_product(form, dy.t().contiguous(), x.t().contiguous())   # the sum over the rows along memory
    if dy.shape[0] >= NATIVE_GEMM_MIN_ROWS and N.gemm_ok(dy, x, True, True):
        return N.gemm(dy, x, True, True)
    return dy.t() @ x


class _LinearFn(torch.autograd.Function):
    """nn.Linear on (rows, in) inputs with the three products on the library's kernel."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.form = _PRODUCT_FORM
        return _mm_forward(x, w, b, ctx.form)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _mm_dgrad(dy, w, ctx.form) if ctx.needs_input_grad[0] else None
        dw = _mm_wgrad(dy, x, ctx.form) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from . import _native as N
            db = N.sum_rows(dy)
        return dx, dw, db


def _linear(x, w, b=None):
    """F.linear for 2-D inputs; many-row fp32 inputs on a HIP device take the library's products."""
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and (
            x.shape[0] >= NATIVE_GEMM_MIN_ROWS or _odd_width_few_rows(x.shape[0], w.shape[0])):
        return _LinearFn.apply(x if x.stride(1) == 1 else x.contiguous(), w, b)
    return F.linear(x, w, b)


class _Linear(nn.Linear):
    """nn.Linear (same parameters and state-dict keys) whose many-row GPU evaluations use ``_linear``."""

    def forward(self, x):
        return _linear(x, self.weight, self.bias) if x.dim() == 2 else super().forward(x)


class _LinearNormActFn(torch.autograd.Function):
    """nn.Linear -> nn.LayerNorm(width) -> nn.SiLU on (rows, in) inputs: the normalisation + activation and
    their backward (with the Linear's bias gradient as a by-product of the parameter-gradient pass) are the
    library's row kernels, the three products its fp32 MFMA kernel (the backend's routine for few rows)."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, eps):
        from . import _native as N
        ctx.form = _PRODUCT_FORM
        pre = _mm_forward(x, w, b, ctx.form)
        y, mean, rstd = N.ln_silu_forward(pre, gamma, beta, eps)
        ctx.save_for_backward(x, w, pre, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _native as N
        x, w, pre, gamma, beta, mean, rstd = ctx.saved_tensors
        dpre, dparams, db = N.ln_silu_backward(pre, dy.contiguous(), gamma, beta, mean, rstd, want_dx_sum=True)
        dx = _mm_dgrad(dpre, w, ctx.form) if ctx.needs_input_grad[0] else None
        return dx, _mm_wgrad(dpre, x, ctx.form), db, dparams[0], dparams[1], None


class _MLP(nn.Sequential):
    """nn.Sequential of (Linear, LayerNorm, SiLU) triples; on the GPU each triple runs as one fused block."""

    fuse = os.environ.get("DRM_NO_MLP_FUSION", "0") != "1"

    def forward(self, x):
        mods = list(self)
        if not (self.fuse and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 1):
            return super().forward(x)
        i = 0
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, nn.Linear) and m.bias is not None and i + 2 < len(mods) and isinstance(mods[i + 1], nn.LayerNorm)
                    and isinstance(mods[i + 2], nn.SiLU) and tuple(mods[i + 1].normalized_shape) == (m.out_features,)
                    and m.out_features % 4 == 0 and m.out_features <= 2048):
                ln = mods[i + 1]
                x = _LinearNormActFn.apply(x.contiguous(), m.weight, m.bias, ln.weight, ln.bias, ln.eps)
                i += 3
            else:
                x = m(x)
                i += 1
        return x


def _norm_act_mlp(in_dim: int, width: int, depth: int) -> nn.Sequential:
    """depth x (Linear, LayerNorm, SiLU), laid out so that state-dict keys are mlp.{0,1,3,4,...}."""
    mods = []
    for i in range(depth):
        mods += [_Linear(in_dim if i == 0 else width, width), nn.LayerNorm(width), nn.SiLU()]
    return _MLP(*mods)


def hard_sample(log_probs: torch.Tensor, expo: Optional[torch.Tensor] = None):
    """Straight-through Gumbel-softmax draw (tau = 1, hard): returns (z, y_soft, idx).

    ``expo`` is the Exp(1) tensor torch's gumbel_softmax would draw; when None it is drawn here
    from the global generator exactly as ``F.gumbel_softmax`` does.
    """
    if expo is None:
        expo = torch.empty_like(log_probs, memory_format=torch.legacy_contiguous_format).exponential_()
    y = torch.softmax(log_probs - expo.log(), dim=-1)
    idx = y.max(dim=-1, keepdim=True)[1]
    hard = torch.zeros_like(y).scatter_(-1, idx, 1.0)
    return hard - y.detach() + y, y, idx.squeeze(-1)


class SequenceModel(nn.Module):
    """h_t = GRU(Linear(cat[z_{t-1}, emb(a_{t-1})]), h_{t-1}) with episode-boundary resets."""

    def __init__(self, action_size: int, num_latents: int = 32, classes_per_latent: int = 32, hidden_size: int = 512):
        super().__init__()
        self.a_emb = nn.Embedding(action_size, 64)
        self.linear = nn.Linear(num_latents * classes_per_latent + 64, hidden_size)
        self.norm = nn.LayerNorm(hidden_size)  # present in the reference's state dict, never applied
        self.act = nn.SiLU()
        self.gru = nn.GRUCell(hidden_size, hidden_size)

    def forward(self, h_prev, z_prev, a_prev_idx, c_prev=None, expo_reset=None):
        emb = self.a_emb(a_prev_idx)
        if c_prev is not None:
            keep = c_prev.view(-1, 1, 1)
            fresh, _, _ = hard_sample(torch.full_like(z_prev, 1.0 / z_prev.size(-1)), expo_reset)
            h_prev = h_prev * c_prev
            z_prev = z_prev * keep + (1.0 - keep) * fresh
            emb = emb * c_prev
        x = self.linear(torch.cat([z_prev.flatten(1), emb], dim=-1))
        return self.gru(x, h_prev)


class _ConvBiasFn(torch.autograd.Function):
    """A (transposed) convolution whose bias gradient is reduced over the contiguous H*W axis first.

    Autograd's own bias gradient is one strided reduction over (N, H, W): 284 us for a (1024, 32, 32, 32)
    gradient on MI355X against 33 us for the two-step form, twice per world-model update.  Input and
    weight gradients come from the same backend routine autograd would call."""

    native_convt = os.environ.get("DRM_NO_NATIVE_CONVT", "0") != "1"

    @staticmethod
    def forward(ctx, x, weight, bias, transposed, stride, padding):
        ctx.save_for_backward(x, weight)
        ctx.conf = (transposed, stride, padding)
        ctx.direct = (transposed and weight.shape[1] <= 4 and tuple(weight.shape[2:]) == (4, 4) and tuple(stride) == (2, 2)
                      and tuple(padding) == (1, 1) and x.dtype == torch.float32 and _ConvBiasFn.native_convt)
        ctx.img = False
        if ctx.direct:
            # few output channels: a stream, served by the library's direct kernels (bias included) in either memory
            # format of the input
            from . import _native as N
            ctx.img = N._is_channels_last(x) and not x.is_contiguous() and N.conv_img_ok(weight.shape[0], weight.shape[1], up=True)
            if ctx.img:
                return N.conv_img_up(x, weight.contiguous(), bias)
            if x.is_contiguous():
                return N.convt_k4s2_forward(x, weight.contiguous(), bias)
        if transposed:
            return F.conv_transpose2d(x, weight, bias, stride=stride, padding=padding)
        return F.conv2d(x, weight, bias, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        transposed, stride, padding = ctx.conf
        g = g.contiguous()
        if ctx.direct:
            # input gradient from the library's stream kernel, in the input's own memory format; the weight gradient
            # (a reduction over all pixels) from the backend's routine
            from . import _native as N
            gx = N.convt_k4s2_input_grad(g, weight.contiguous(), N._is_channels_last(x)) if ctx.needs_input_grad[0] else None
            if ctx.img:
                gw = N.conv_img_wgrad(x, g)       # channels-last input: the reduction over all pixels on the library's kernel
            else:
                _, gw, _ = torch.ops.aten.convolution_backward(g, x, weight, None, list(stride), list(padding), [1, 1],
                                                               transposed, [0, 0], 1, [False, True, False])
        else:
            gx, gw, _ = torch.ops.aten.convolution_backward(g, x, weight, None, list(stride), list(padding), [1, 1], transposed,
                                                            [0, 0], 1, [ctx.needs_input_grad[0], True, False])
        gb = g.view(g.shape[0], g.shape[1], -1).sum(2).sum(0)
        return gx, gw, gb, None, None, None


def _conv_forward(mod, x, transposed: bool):
    hw = x.shape[-1] * x.shape[-2] * (4 if transposed else 1) // (1 if transposed else 4)   # output H*W at stride 2
    if x.is_cuda and torch.is_grad_enabled() and mod.bias is not None and mod.bias.requires_grad and hw >= 256:
        return _ConvBiasFn.apply(x, mod.weight, mod.bias, transposed, mod.stride, mod.padding)
    return None


class _Conv2d(nn.Conv2d):
    def forward(self, x):
        y = _conv_forward(self, x, False)
        return y if y is not None else super().forward(x)


class _ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x):
        y = _conv_forward(self, x, True)
        return y if y is not None else super().forward(x)


class _LayerNormSiLUFn(torch.autograd.Function):
    """nn.LayerNorm([C, H, W]) followed by nn.SiLU as one HIP kernel forward and two backward; with
    ``cbias`` the bias of the convolution that produced ``x`` is added inside the same kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, cbias=None):
        from . import _native as N
        if not N._is_channels_last(x):
            x = x.contiguous()
        y, mean, rstd = N.ln_silu_forward(x, weight, bias, eps, cbias)
        ctx.save_for_backward(x, weight, bias, mean, rstd, cbias)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _native as N
        x, weight, bias, mean, rstd, cbias = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last) if N._is_channels_last(x) else dy.contiguous()
        if cbias is None:
            dx, dparams = N.ln_silu_backward(x, dy, weight, bias, mean, rstd)
            return dx, dparams[0].view_as(weight), dparams[1].view_as(bias), None, None
        dx, dparams, dcb = N.ln_silu_backward(x, dy, weight, bias, mean, rstd, cbias)
        return dx, dparams[0].view_as(weight), dparams[1].view_as(bias), None, dcb


class _Conv4s2Fn(torch.autograd.Function):
    """nn.Conv2d / nn.ConvTranspose2d (kernel 4, stride 2, padding 1, no bias) of the towers on the library's
    implicit-product kernels: forward, input gradient and weight gradient, activations in channels-last memory.
    ``up``: the transposed convolution (small -> big resolution); its weight (in, out, 4, 4) has the small-resolution
    side first, as the convolution's (out, in, 4, 4) has, so both share the three kernels."""

    @staticmethod
    def forward(ctx, x, weight, up):
        from . import _native as N
        ctx.save_for_backward(x, weight)
        ctx.up = up
        ctx.bf16 = _PRODUCT_FORM == "bf16"       # inside WorldModel.loss of a model whose update takes bf16 matrix inputs
        return N.conv4s2_up(x, weight, ctx.bf16) if up else N.conv4s2_down(x, weight, ctx.bf16)

    @staticmethod
    def backward(ctx, g):
        from . import _native as N
        x, weight = ctx.saved_tensors
        gx = gw = None
        if ctx.up:
            if ctx.needs_input_grad[0]:
                gx = N.conv4s2_down(g, weight, ctx.bf16)
            if ctx.needs_input_grad[1]:
                gw = N.conv4s2_wgrad(x, g, ctx.bf16)
        else:
            if ctx.needs_input_grad[0]:
                gx = N.conv4s2_up(g, weight, ctx.bf16)
            if ctx.needs_input_grad[1]:
                gw = N.conv4s2_wgrad(g, x, ctx.bf16)
        return gx, gw, None


class _ConvImgDownFn(torch.autograd.Function):
    """nn.Conv2d(Cimg <= 4 -> Cs, kernel 4, stride 2, padding 1, no bias) on an image: the first encoder layer on the
    library's kernels (forward and weight gradient); the result is in channels-last memory."""

    @staticmethod
    def forward(ctx, x, weight):
        from . import _native as N
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        return N.conv_img_down(x, weight.contiguous())

    @staticmethod
    def backward(ctx, g):
        from . import _native as N
        x, weight = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[1]:
            gw = N.conv_img_wgrad(g, x)
        if ctx.needs_input_grad[0]:        # an image that requires a gradient: not a training path; the backend's routine
            gx = torch.ops.aten.convolution_backward(g, x, weight, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        return gx, gw


def _image_conv_ok(mod, x) -> bool:
    """The first encoder layer (an image with at most 4 channels in, a multiple of 32 channels out) on the library's
    image-side kernels."""
    if not (isinstance(mod, nn.Conv2d) and not isinstance(mod, nn.ConvTranspose2d)):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] >= NATIVE_CONV_MIN_IMAGES
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0):
        return False
    if not (tuple(mod.kernel_size) == (4, 4) and tuple(mod.stride) == (2, 2) and tuple(mod.padding) == (1, 1)
            and tuple(mod.dilation) == (1, 1) and mod.groups == 1 and mod.padding_mode == "zeros"):
        return False
    from . import _native as N
    return N.conv_img_ok(mod.out_channels, mod.in_channels)


# the towers' stride-2 4x4 layers run on the library's kernels from this many images on (DRM_NO_NATIVE_CONV=1: never)
NATIVE_CONV_MIN_IMAGES = 10 ** 9 if os.environ.get("DRM_NO_NATIVE_CONV", "0") == "1" else int(os.environ.get("DRM_NATIVE_CONV_MIN_IMAGES", "16"))


def _native_conv_ok(mod, x) -> bool:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] >= NATIVE_CONV_MIN_IMAGES):
        return False
    if not (tuple(mod.kernel_size) == (4, 4) and tuple(mod.stride) == (2, 2) and tuple(mod.padding) == (1, 1)
            and tuple(mod.dilation) == (1, 1) and mod.groups == 1):
        return False
    up = isinstance(mod, nn.ConvTranspose2d)
    if up and tuple(mod.output_padding) != (0, 0):
        return False
    from . import _native as N
    cs, cb = mod.weight.shape[0], mod.weight.shape[1]      # small- and big-resolution channel counts, either module
    return all(N.conv4s2_ok(cs, cb, m) for m in ("down", "up", "wgrad"))


def _conv_no_bias(mod, x, image_conv: bool = False):
    if image_conv and _image_conv_ok(mod, x):
        return _ConvImgDownFn.apply(x, mod.weight)
    if _native_conv_ok(mod, x):
        return _Conv4s2Fn.apply(x, mod.weight, isinstance(mod, nn.ConvTranspose2d))
    if isinstance(mod, nn.ConvTranspose2d):
        return F.conv_transpose2d(x, mod.weight, None, mod.stride, mod.padding, mod.output_padding, mod.groups, mod.dilation)
    return F.conv2d(x, mod.weight, None, mod.stride, mod.padding, mod.dilation, mod.groups)


class _Tower(nn.Sequential):
    """nn.Sequential whose (LayerNorm over the whole sample, SiLU) pairs run fused on the GPU, together
    with the bias of the convolution in front of them; the modules, and therefore the state-dict keys,
    are those of the plain Sequential."""

    fuse = os.environ.get("DRM_NO_LN_FUSION", "0") != "1"
    # activations of the tower in the channels-last memory format (and, through the flat parameter buffers, the
    # convolution kernels too): the layout the library's convolution kernels read and write, so no layout copies
    # between the stages; parameters and results keep their logical (N, C, H, W) meaning.  Measured on the 16 x 64
    # training update: 11.13 ms against 11.66 ms with plain NCHW tensors (DRM_TOWER_CHANNELS_LAST=0).
    channels_last = os.environ.get("DRM_TOWER_CHANNELS_LAST", "1") == "1"

    def _inner_convs_native(self, x) -> bool:
        """Channels-last activations are the layout of the library's own convolution kernels; a tower takes it only when
        every convolution apart from the image-side one (the first of an encoder, the last of a decoder) runs on them.
        Towers with other channel counts keep plain NCHW tensors, so the backend never receives channels-last tensors
        with a handful of channels (a combination that only reduced-size test models produce)."""
        convs = [m for m in self if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))]
        if len(convs) < 2:
            return False
        inner = convs[:-1] if isinstance(convs[-1], nn.ConvTranspose2d) else convs[1:]
        return all(_native_conv_ok(m, x) for m in inner)

    @staticmethod
    def _fusable(m, nxt, shape, x):
        return (isinstance(m, nn.LayerNorm) and isinstance(nxt, nn.SiLU) and x.dtype == torch.float32 and len(shape) == 4
                and tuple(m.normalized_shape) == tuple(shape[1:]) and m.weight.numel() % 4 == 0 and m.weight.numel() <= 32768)

    def forward(self, x):
        mods = list(self)
        i = 0
        if not self.fuse or not x.is_cuda:
            return super().forward(x)
        cl = self.channels_last and x.dim() == 4 and x.dtype == torch.float32 and x.shape[0] >= 64 and self._inner_convs_native(x)
        # an encoder's image goes to the image-side kernel as it is (NCHW in, channels-last out)
        if cl and not (len(mods) > 0 and _image_conv_ok(mods[0], x)):
            x = x.contiguous(memory_format=torch.channels_last)
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, (_Conv2d, _ConvTranspose2d)) and m.bias is not None and i + 2 < len(mods)
                    and isinstance(mods[i + 1], nn.LayerNorm) and x.dim() == 4 and x.dtype == torch.float32):
                ln = mods[i + 1]
                shape = (x.shape[0],) + tuple(ln.normalized_shape)
                hw = shape[2] * shape[3] if len(shape) == 4 else 0
                if self._fusable(ln, mods[i + 2], shape, x) and hw % 4 == 0 and shape[1] == m.out_channels:
                    y = _conv_no_bias(m, x, image_conv=cl and i == 0)
                    if shape[1] % 4 != 0:
                        y = y.contiguous()     # the channels-last kernels take channel counts that are multiples of 4
                    x = _LayerNormSiLUFn.apply(y, ln.weight, ln.bias, ln.eps, m.bias)
                    i += 3
                    continue
            if i + 1 < len(mods) and self._fusable(m, mods[i + 1], tuple(x.shape), x):
                if x.shape[1] % 4 != 0:
                    x = x.contiguous()
                x = _LayerNormSiLUFn.apply(x, m.weight, m.bias, m.eps)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


def _conv_tower(channels, spatial, transpose: bool):
    """Conv(k4,s2,p1) + LayerNorm[C,H,W] + SiLU per stage; spatial[i] is the OUTPUT size of stage i."""
    mods = []
    conv = _ConvTranspose2d if transpose else _Conv2d
    for i in range(len(channels) - 1):
        mods += [conv(channels[i], channels[i + 1], 4, stride=2, padding=1),
                 nn.LayerNorm([channels[i + 1], spatial[i][0], spatial[i][1]]), nn.SiLU()]
    return mods


class Encoder(nn.Module):
    """Posterior logits from cat[conv(x_t), h_t]."""

    def __init__(self, obs_shape, num_latents=32, classes_per_latent=32, hidden_size=512, base_cnn_channels=32):
        super().__init__()
        C, H, W = obs_shape
        assert H % 16 == 0 and W % 16 == 0, "H and W must be divisible by 16"
        self.C, self.H, self.W = C, H, W
        self.num_latents, self.classes_per_latent = num_latents, classes_per_latent
        c = base_cnn_channels
        sizes = [(H >> s, W >> s) for s in (1, 2, 3, 4)]
        self.conv = _Tower(*_conv_tower([C, c, 2 * c, 4 * c, 8 * c], sizes, False), nn.Flatten())
        self.conv_out_size = 8 * c * (H // 16) * (W // 16)
        self.fc = nn.Sequential(_Linear(self.conv_out_size + hidden_size, num_latents * classes_per_latent))

    def forward(self, x_t, h_t):
        B, C, H, W = x_t.shape
        assert (C, H, W) == (self.C, self.H, self.W), f"Got {(C, H, W)}, expected {(self.C, self.H, self.W)}"
        feat = torch.cat([self.conv(x_t), h_t], dim=-1)
        return self.fc(feat).view(-1, self.num_latents, self.classes_per_latent)


class DynamicsPredictor(nn.Module):
    """Prior logits from h_t."""

    def __init__(self, hidden_size=512, num_latents=32, classes_per_latent=32, mlp_hidden_units=512, mlp_layers=3):
        super().__init__()
        self.mlp = _norm_act_mlp(hidden_size, mlp_hidden_units, mlp_layers)
        self.head = _Linear(mlp_hidden_units, num_latents * classes_per_latent)
        self.num_latents, self.classes_per_latent = num_latents, classes_per_latent

    def forward(self, h_t):
        return self.head(self.mlp(h_t)).view(-1, self.num_latents, self.classes_per_latent)


class _ScalarHead(nn.Module):
    def __init__(self, width_h, num_latents, classes_per_latent, mlp_hidden_units, mlp_layers):
        super().__init__()
        self.mlp = _norm_act_mlp(width_h + num_latents * classes_per_latent, mlp_hidden_units, mlp_layers)
        self.head = _Linear(mlp_hidden_units, 1)
        self.num_latents, self.classes_per_latent = num_latents, classes_per_latent

    def forward(self, h_t, z_t):
        return self.head(self.mlp(torch.cat([h_t, z_t.reshape(h_t.size(0), -1)], dim=-1)))


class RewardPredictor(_ScalarHead):
    def __init__(self, hidden_size=512, num_latents=32, classes_per_latent=32, mlp_hidden_units=512, mlp_layers=3):
        super().__init__(hidden_size, num_latents, classes_per_latent, mlp_hidden_units, mlp_layers)


class ContinuePredictor(_ScalarHead):
    def __init__(self, h_dim=512, num_latents=32, classes_per_latent=32, mlp_hidden_units=512, mlp_layers=3):
        super().__init__(h_dim, num_latents, classes_per_latent, mlp_hidden_units, mlp_layers)


class Decoder(nn.Module):
    """Image reconstruction from cat[h_t, z_t]."""

    def __init__(self, obs_shape, num_latents=32, classes_per_latent=32, hidden_size=512, base_cnn_channels=32):
        super().__init__()
        C, H, W = obs_shape
        assert H % 16 == 0 and W % 16 == 0, "H and W must be divisible by 16"
        self.C, self.H, self.W = C, H, W
        self.base_h, self.base_w = H // 16, W // 16
        self.cnn_multiplier = c = base_cnn_channels
        self.fc = nn.Sequential(_Linear(hidden_size + num_latents * classes_per_latent, 8 * c * self.base_h * self.base_w),
                                nn.SiLU())
        sizes = [(self.base_h << s, self.base_w << s) for s in (1, 2, 3)]
        self.deconv = _Tower(*_conv_tower([8 * c, 4 * c, 2 * c, c], sizes, True),
                                    _ConvTranspose2d(c, C, 4, stride=2, padding=1), nn.Sigmoid())

    def forward(self, h_t, z_t):
        B = h_t.size(0)
        seed = self.fc(torch.cat([h_t, z_t.reshape(B, -1)], dim=-1))
        img = self.deconv(seed.view(B, 8 * self.cnn_multiplier, self.base_h, self.base_w))
        assert img.shape == (B, self.C, self.H, self.W)
        return img


class WorldModel(nn.Module):
    def __init__(self, obs_shape, action_size: int, num_latents: int = 32, classes_per_latent: int = 32,
                 hidden_size: int = 512, base_cnn_channels: int = 32, mlp_hidden_units: int = 512,
                 free_bits: float = 1.0, beta_pred: float = 1.0, beta_dyn: float = 0.5, beta_rep: float = 0.1,
                 unimix_eps: float = 0.01):
        super().__init__()
        self.obs_shape, self.action_size, self.hidden_size = obs_shape, action_size, hidden_size
        self.num_latents, self.classes_per_latent = num_latents, classes_per_latent
        self.free_bits, self.unimix_eps = free_bits, unimix_eps
        self.beta_pred, self.beta_dyn, self.beta_rep = beta_pred, beta_dyn, beta_rep
        lk = dict(num_latents=num_latents, classes_per_latent=classes_per_latent)
        self.seq = SequenceModel(action_size=action_size, hidden_size=hidden_size, **lk)
        self.enc = Encoder(obs_shape=obs_shape, hidden_size=hidden_size, base_cnn_channels=base_cnn_channels, **lk)
        self.dyn = DynamicsPredictor(hidden_size=hidden_size, mlp_hidden_units=mlp_hidden_units, mlp_layers=3, **lk)
        self.rew = RewardPredictor(hidden_size=hidden_size, mlp_hidden_units=mlp_hidden_units, mlp_layers=3, **lk)
        self.cont = ContinuePredictor(h_dim=hidden_size, mlp_hidden_units=mlp_hidden_units, mlp_layers=3, **lk)
        self.dec = Decoder(obs_shape=obs_shape, hidden_size=hidden_size, base_cnn_channels=base_cnn_channels, **lk)
        for head in (self.rew.head, self.cont.head):  # zero-init heads, lib/world_model.py:398-401
            nn.init.zeros_(head.weight)
            nn.init.zeros_(head.bias)

    # ------------------------------------------------------------------ single steps (any device, eager)
    @torch.no_grad()
    def init_state(self, batch_size: int, device=None, dtype=None, expo: Optional[torch.Tensor] = None) -> WorldModelState:
        h0 = torch.zeros(batch_size, self.hidden_size, device=device, dtype=dtype)
        flat = torch.full((batch_size, self.num_latents, self.classes_per_latent), 1.0 / self.classes_per_latent,
                          device=device, dtype=dtype)
        z0, _, _ = hard_sample(flat, expo)
        return WorldModelState(h=h0, z=z0)

    def _sample_z(self, logits: torch.Tensor, expo: Optional[torch.Tensor] = None) -> torch.Tensor:
        return hard_sample(log_unimix(logits, self.unimix_eps, dim=-1), expo)[0]

    def step(self, state_prev: WorldModelState, a_prev_idx: torch.Tensor, x_cur: Optional[torch.Tensor] = None,
             c_prev: Optional[torch.Tensor] = None) -> Tuple[WorldModelState, Dict[str, Any]]:
        """One RSSM step: posterior when ``x_cur`` is given, prior otherwise.  Same outputs as the
        reference's step (lib/world_model.py:415-457), heads and decoder included."""
        h = self.seq(state_prev.h, state_prev.z, a_prev_idx, c_prev)
        post = self.enc(x_cur, h) if x_cur is not None else None
        prior = self.dyn(h)
        z = self._sample_z(post if post is not None else prior)
        info = {"prior_logits": prior, "post_logits": post, "r_hat": self.rew(h, z), "c_hat": self.cont(h, z),
                "x_hat": self.dec(h, z)}
        return WorldModelState(h=h, z=z), info

    @torch.no_grad()
    def act_step(self, state_prev: WorldModelState, a_prev_idx: torch.Tensor, x_cur: torch.Tensor,
                 c_prev: Optional[torch.Tensor] = None) -> WorldModelState:
        """Posterior state after one observed frame, without the heads and the decoder.

        The acting loop keeps only the state of ``step`` (lib/trainer.py:126-130 discards the info dict), so
        the prior, reward, continue and decoder evaluations are skipped here.  The generator is consumed as
        in ``step``: the reset draw (only when ``c_prev`` is given), then the sample draw, (B,L,K) each.
        On the GPU the recurrence runs through the fused scan kernels with T = 1.
        """
        from .observe import observe_scan
        B = x_cur.shape[0]
        L, K = self.num_latents, self.classes_per_latent
        dev, dt = x_cur.device, x_cur.dtype
        expo_z = torch.empty(1, B, L, K, device=dev, dtype=dt)
        if c_prev is None:
            expo_reset, c = expo_z.exponential_(), torch.ones(B, 1, device=dev, dtype=dt)   # c = 1: the reset draw is not used
        else:
            expo_reset, c = torch.empty(1, B, L, K, device=dev, dtype=dt).exponential_(), c_prev.reshape(B, 1).to(dt)
            expo_z.exponential_()
        fc = self.enc.fc[0]
        n_conv = self.enc.conv_out_size
        e_part = F.linear(self.enc.conv(x_cur), fc.weight[:, :n_conv], fc.bias).view(B, 1, L * K)
        return observe_scan(self, e_part, a_prev_idx.reshape(B, 1), c, expo_reset, expo_z, state_prev,
                            engine_attr="_observe_engine_act")[3]

    # ------------------------------------------------------------------ loss terms
    def _kl_loss(self, q_logits: torch.Tensor, p_logits: torch.Tensor) -> torch.Tensor:
        lq = log_unimix(q_logits, self.unimix_eps, dim=-1)
        lp = log_unimix(p_logits, self.unimix_eps, dim=-1)
        per_latent = (lq.exp() * (lq - lp)).sum(dim=-1)
        return per_latent.clamp_min(self.free_bits / float(self.num_latents)).sum(dim=-1)

    @staticmethod
    def _prediction_loss(x_true, x_hat, r_true, r_hat, c_true, c_hat) -> torch.Tensor:
        image = 0.5 * (x_hat - x_true).pow(2).mean(dim=(-3, -2, -1))
        reward = 0.5 * (r_hat - symlog(r_true)).pow(2).mean(dim=-1)
        cont = F.binary_cross_entropy_with_logits(c_hat, c_true, reduction="none").squeeze(-1)
        return image + reward + cont

    def _dynamics_loss(self, post_logits, prior_logits) -> torch.Tensor:
        return self._kl_loss(post_logits.detach(), prior_logits)

    def _representation_loss(self, post_logits, prior_logits) -> torch.Tensor:
        return self._kl_loss(post_logits, prior_logits.detach())

    # ------------------------------------------------------------------ observe: sequence loss
    def observe(self, obs, actions, continues, init_state: Optional[WorldModelState] = None,
                noise: Optional[Dict[str, torch.Tensor]] = None):
        """Posterior rollout over a replayed (B, T) batch.

        Returns (h (B,T,D), z (B,T,L,K) straight-through one-hot, post_logits (B,T,L,K), last state).
        ``noise`` holds the Exp(1) draws: ``init`` (B,L,K) (unused when init_state is given),
        ``reset`` and ``z`` (T,B,L,K); drawn from the global generator when None.
        """
        from .observe import observe_scan
        B, T = obs.shape[:2]
        L, K = self.num_latents, self.classes_per_latent
        dev, dt = obs.device, obs.dtype
        if noise is None:
            noise = {"reset": torch.empty(T, B, L, K, device=dev, dtype=dt).exponential_(),
                     "z": torch.empty(T, B, L, K, device=dev, dtype=dt).exponential_()}
            if init_state is None:
                noise["init"] = torch.empty(B, L, K, device=dev, dtype=dt).exponential_()
        state = init_state if init_state is not None else self.init_state(B, device=dev, dtype=dt, expo=noise["init"])
        a_prev = torch.roll(actions, 1, dims=1)
        a_prev[:, 0] = 0
        c = continues.to(dt)
        c_prev = torch.cat([torch.ones(B, 1, device=dev, dtype=dt), c[:, :-1]], dim=1)  # (B,T)
        fc = self.enc.fc[0]
        n_conv = self.enc.conv_out_size
        feat = self.enc.conv(obs.reshape(B * T, *obs.shape[2:]))
        e_part = _linear(feat, fc.weight[:, :n_conv], fc.bias).view(B, T, L * K)
        return observe_scan(self, e_part, a_prev, c_prev, noise["reset"], noise["z"], state)

    def loss(self, obs, actions, rewards, continues, init_state: Optional[WorldModelState] = None,
             noise: Optional[Dict[str, torch.Tensor]] = None):
        """World-model loss over a (B, T) batch; same value and gradients as lib/world_model.py:530-603."""
        B, T = obs.shape[:2]
        L, K = self.num_latents, self.classes_per_latent
        # a wide model whose scan takes bf16 matrix inputs: the towers' convolutions and the large batched Linear products of
        # the heads as well
        with _bf16_products(getattr(self, "observe_precision", "f32") == "bf16" and self.hidden_size > 512):
            h, z, post, last = self.observe(obs, actions, continues, init_state, noise)
            hf, zf = h.reshape(B * T, -1), z.reshape(B * T, L, K)
            return self._loss_from_states(obs, rewards, continues, h, z, post, last, hf, zf)

    def _loss_from_states(self, obs, rewards, continues, h, z, post, last, hf, zf):
        B, T = obs.shape[:2]
        L, K = self.num_latents, self.classes_per_latent
        prior = self.dyn(hf)
        postf = post.reshape(B * T, L, K)
        x_true, x_hat = obs.reshape(B * T, *obs.shape[2:]), self.dec(hf, zf)
        from . import losses as FL
        if FL.fused_ok(postf, K) and x_hat.numel() // (B * T) % 4 == 0:
            # the same terms from the library's row kernels (one launch each way instead of the eager chains)
            image = FL.mse_rows(x_hat, x_true)
            reward = 0.5 * (self.rew(hf, zf) - symlog(rewards.reshape(B * T, 1))).pow(2).mean(dim=-1)
            cont = F.binary_cross_entropy_with_logits(self.cont(hf, zf), continues.to(obs.dtype).reshape(B * T, 1),
                                                      reduction="none").squeeze(-1)
            pred = (image + reward + cont).mean()
            dyn_rows, rep_rows = FL.kl_pair(postf, prior, self.unimix_eps, self.free_bits / float(L))
            dyn, rep = dyn_rows.mean(), rep_rows.mean()
        else:
            pred = self._prediction_loss(x_true, x_hat, rewards.reshape(B * T, 1), self.rew(hf, zf),
                                         continues.to(obs.dtype).reshape(B * T, 1), self.cont(hf, zf)).mean()
            dyn = self._dynamics_loss(postf, prior).mean()
            rep = self._representation_loss(postf, prior).mean()
        total = self.beta_pred * pred + self.beta_dyn * dyn + self.beta_rep * rep
        return total, {"total_loss": total, "pred_loss": pred, "dyn_loss": dyn, "rep_loss": rep, "state": last,
                       "h": h, "z": z, "post_logits": post, "prior_logits": prior.view(B, T, L, K)}
