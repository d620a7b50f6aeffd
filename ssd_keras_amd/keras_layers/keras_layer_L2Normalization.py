"""`L2Normalization` -- drop-in for keras_layers/keras_layer_L2Normalization.py:25-70 as a torch module.

`K.l2_normalize(x, axis=channels) * gamma` with TensorFlow's epsilon placement:
x * rsqrt(max(sum_c x^2, 1e-12)), gamma a learnable per-channel scale initialised to 20.
Operates on NCHW tensors (channels_last memory keeps the reduction contiguous).
"""
from __future__ import annotations

import torch
from torch import nn


class _L2NormFn(torch.autograd.Function):
    """float32 / bf16 maps on a GPU, with or without gradients: one libssdhip pass forward (the pixel's inverse norm is kept), one
    backward (csrc/ssdhip_train.hip) -- instead of seven elementwise / reduction kernels each way."""

    @staticmethod
    def forward(ctx, x, gamma):
        from .. import _native as nat
        y, inv = nat.l2_normalize_fwd(x, gamma, want_inv=True)
        ctx.save_for_backward(x, gamma, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _native as nat
        x, gamma, inv = ctx.saved_tensors
        dx, dgamma = nat.l2_normalize_bwd(x, dy, gamma, inv)
        return dx, dgamma.to(gamma.dtype)


class L2Normalization(nn.Module):
    def __init__(self, gamma_init=20, n_channels=None, **kwargs):
        super().__init__()
        self.gamma_init = gamma_init
        self.fused_inference = True
        self.name = kwargs.get('name')
        self.gamma = nn.Parameter(torch.full((n_channels,), float(gamma_init))) if n_channels else None

    def build(self, input_shape, device=None):
        """Create gamma (reference :54-59: one value per channel, initialised to `gamma_init`).  `input_shape`: the number of channels,
        or the shape of the NCHW tensors this module takes (channels on axis 1; the Keras layer reads axis 3 of NHWC)."""
        n_channels = int(input_shape) if isinstance(input_shape, int) else int(tuple(input_shape)[1])
        self.gamma = nn.Parameter(torch.full((n_channels,), float(self.gamma_init), device=device))

    def forward(self, x):
        if self.gamma is None:
            self.build(x.shape[1], x.device)
        if (self.fused_inference and x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled()
                and x.shape[1] % 8 == 0):
            from .. import _native as nat          # one pass in libssdhip (csrc/ssdhip_layers.hip) instead of seven kernels
            return nat.l2_normalize(x, self.gamma_float32())
        if self.fused_inference and x.is_cuda and x.dim() == 4:
            from .. import _native as nat
            if nat.l2_normalize_supported(x):               # the float32 model, and the training step in either dtype
                return _L2NormFn.apply(x, self.gamma)
        xf = x.float()
        inv = torch.rsqrt(torch.clamp_min((xf * xf).sum(dim=1, keepdim=True), 1e-12))
        return (xf * inv * self.gamma.view(1, -1, 1, 1)).to(x.dtype)

    def gamma_float32(self):
        """gamma as the float32 tensor the kernels read: itself for a float32 model, a cached copy for a bf16 one (converted once, not
        once per step; refreshed in its own storage when gamma changed -- a captured graph may be reading it)."""
        g = self.gamma
        if g.dtype == torch.float32:
            return g
        key = (g.data_ptr(), g._version)
        hit = self.__dict__.get("_gamma32")
        if hit is not None and hit[0] != key and hit[1].device == g.device and hit[1].shape == g.shape \
                and not torch.cuda.is_current_stream_capturing():
            self.refresh_cached_gamma()
            hit = self.__dict__["_gamma32"]
        elif hit is None or hit[0] != key:
            hit = (key, g.detach().float().contiguous())
            if not torch.cuda.is_current_stream_capturing():
                self.__dict__["_gamma32"] = hit
        return hit[1]

    def refresh_cached_gamma(self):
        """Bring the cached float32 copy of gamma up to date IN ITS OWN STORAGE (a captured HIP graph keeps reading it)."""
        hit = self.__dict__.get("_gamma32")
        g = self.gamma
        if hit is None or g is None:
            return
        key = (g.data_ptr(), g._version)
        if hit[0] != key and hit[1].device == g.device and hit[1].shape == g.shape:
            with torch.no_grad():
                hit[1].copy_(g.detach())
            self.__dict__["_gamma32"] = (key, hit[1])

    call = forward                                   # the Keras layer's method name (reference :61)

    def get_config(self):
        return {'gamma_init': self.gamma_init}
