"""`SSDLoss` -- drop-in for the reference's keras_loss_function/keras_ssd_loss.py:22-211.

`SSDLoss(neg_pos_ratio=3, n_neg_min=0, alpha=1.0).compute_loss(y_true, y_pred)` returns the per-batch-item
loss tensor `(batch,)` exactly as the Keras loss callable does (Keras then averages it).  Forward and backward
run in libssdhip.so (`ssdhip_loss_forward` / `ssdhip_loss_backward`); the function is differentiable with
respect to `y_pred` through a `torch.autograd.Function`, so `loss.mean().backward()` drives a PyTorch model the
way `model.compile(loss=ssd_loss.compute_loss)` drives the Keras one.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _native as nat


class _SSDLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_true, y_pred, neg_pos_ratio, n_neg_min, alpha):
        lib = nat.load()
        if not hasattr(lib, 'ssdhip_loss_forward'):
            raise nat.SsdHipError("libssdhip.so was built without the loss kernels")
        y_true = y_true.detach()
        yp = y_pred.detach()
        if y_true.dtype != torch.float32:
            y_true = y_true.float()
        if yp.dtype != torch.float32:
            yp = yp.float()
        y_true, yp = y_true.contiguous(), yp.contiguous()
        nat.require_cuda(y_true, 'y_true')
        nat.require_cuda(yp, 'y_pred')
        if y_true.shape != yp.shape or yp.dim() != 3 or yp.shape[2] < 14:
            raise ValueError("y_true and y_pred must both have shape (batch, #boxes, #classes + 12)")
        B, N, L = yp.shape
        C = L - 12
        dev = yp.device
        loss = torch.empty((B,), dtype=torch.float32, device=dev)
        stats = torch.empty((4,), dtype=torch.float32, device=dev)
        keep = torch.empty((B, N), dtype=torch.uint8, device=dev)
        ws = nat.workspaces.get(dev, 'loss', lib.ssdhip_loss_workspace_bytes(B, N, C))
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            rc = lib.ssdhip_loss_forward(p(y_true), p(yp), B, N, C, int(neg_pos_ratio), int(n_neg_min), float(alpha),
                                         p(loss), p(stats), p(keep), p(ws), ws.numel(), nat.current_stream_ptr(dev))
        nat.check(rc, 'ssdhip_loss_forward')
        ctx.save_for_backward(y_true, yp, keep, stats)
        ctx.alpha = float(alpha)
        ctx.in_dtype = y_pred.dtype
        ctx.mark_non_differentiable(stats, keep)
        return loss, stats, keep

    @staticmethod
    def backward(ctx, grad_loss, _gs, _gk):
        y_true, yp, keep, stats = ctx.saved_tensors
        lib = nat.load()
        B, N, L = yp.shape
        grad = torch.empty_like(yp)
        go = grad_loss.detach().float().contiguous()
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(yp.device):
            rc = lib.ssdhip_loss_backward(p(y_true), p(yp), p(keep), p(stats), p(go), B, N, L - 12, ctx.alpha, p(grad),
                                          nat.current_stream_ptr(yp.device))
        nat.check(rc, 'ssdhip_loss_backward')
        return None, grad.to(ctx.in_dtype), None, None, None


class SSDLoss:
    '''The SSD loss, see https://arxiv.org/abs/1512.02325 (reference class keras_ssd_loss.py:22-51).'''

    def __init__(self, neg_pos_ratio=3, n_neg_min=0, alpha=1.0):
        self.neg_pos_ratio = neg_pos_ratio
        self.n_neg_min = n_neg_min
        self.alpha = alpha

    def smooth_L1_loss(self, y_true, y_pred):
        '''Smooth L1 loss summed over the last axis (reference :53-75): `0.5 d^2` where `|d| < 1`, `|d| - 0.5` elsewhere.  A plain,
        differentiable tensor expression on whatever device the tensors live on, like the reference's graph ops -- `compute_loss`
        does not call it: the fused kernel evaluates the same expression per anchor.'''
        y_true, y_pred = torch.as_tensor(y_true), torch.as_tensor(y_pred)
        d = y_true - y_pred
        absolute_loss = d.abs()
        return torch.where(absolute_loss < 1.0, 0.5 * d * d, absolute_loss - 0.5).sum(dim=-1)

    def log_loss(self, y_true, y_pred):
        '''Softmax log loss summed over the last axis (reference :77-96): `-sum(y_true * log(max(y_pred, 1e-15)))`; same remarks.'''
        y_true, y_pred = torch.as_tensor(y_true), torch.as_tensor(y_pred)
        return -(y_true * torch.log(torch.clamp_min(y_pred, 1e-15))).sum(dim=-1)

    def compute_loss(self, y_true, y_pred):
        '''Reference :98-211.  `y_true`, `y_pred`: `(batch, #boxes, #classes + 12)` on the GPU (NumPy `y_true`
        is uploaded).  Returns a `(batch,)` float32 tensor; anchors whose class vector is all zero are ignored,
        the last eight columns are never read.'''
        if not torch.is_tensor(y_true):
            y_true = nat.to_device(y_true, device=y_pred.device, dtype=torch.float32)
        loss, _, _ = _SSDLossFn.apply(y_true, y_pred, self.neg_pos_ratio, self.n_neg_min, self.alpha)
        return loss

    def compute_loss_with_stats(self, y_true, y_pred):
        '''As `compute_loss`, also returning `[n_positive, n_neg_losses, k, k-th negative loss]` and the kept-negative mask.'''
        if not torch.is_tensor(y_true):
            y_true = nat.to_device(y_true, device=y_pred.device, dtype=torch.float32)
        return _SSDLossFn.apply(y_true, y_pred, self.neg_pos_ratio, self.n_neg_min, self.alpha)

    __call__ = compute_loss
