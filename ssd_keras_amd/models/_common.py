"""Shared machinery of the SSD model builders (reference models/keras_ssd300.py:200-457 and twins).

The convolutional stack of a bf16 model runs on libssdhip's MFMA kernels (csrc/ssdhip_convh / conv64 / conv /
convimg / chain .hip), dispatched here per layer shape (`SSDModel._pick`), with the graph glue in
csrc/ssdhip_layers.hip and, for training, autograd functions over the same kernels (+ csrc/ssdhip_wgrad /
ssdhip_train .hip); the framework's own convolution (MIOpen) only runs float32 models and the few layer
geometries no kernel here covers.  Around it this module adds what the reference's graph does: in-graph input
normalisation, NHWC-ordered head reshapes so the anchor axis factorises exactly like Keras'
`Reshape((-1, n_classes))`, the resident anchor constant, softmax, the `(B, N, C+12)` prediction layout and the
optional decode layer.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from .. import _native as nat
from ..anchor_math import n_boxes_for
from ..optimizers import _bump_versions
from ..keras_layers.keras_layer_AnchorBoxes import AnchorBoxes
from ..keras_layers.keras_layer_DecodeDetections import DecodeDetections
from ..keras_layers.keras_layer_DecodeDetectionsFast import DecodeDetectionsFast
from ..keras_layers.keras_layer_L2Normalization import L2Normalization


def conv_out(n, k, s=1, p=0, d=1):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def pool_out(n, k, s, p=0, ceil_mode=False):
    if ceil_mode:
        o = -(-(n + 2 * p - k) // s) + 1
        if (o - 1) * s >= n + p:
            o -= 1
        return o
    return (n + 2 * p - k) // s + 1


def he_normal_(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_in', nonlinearity='relu')   # Keras 'he_normal'
            if m.bias is not None:
                nn.init.zeros_(m.bias)


def resolve_anchor_config(n_predictor_layers, min_scale, max_scale, scales, aspect_ratios_global,
                          aspect_ratios_per_layer, two_boxes_for_ar1, steps, offsets, variances):
    """The argument checks and defaults of the builders (keras_ssd300.py:178-237)."""
    if aspect_ratios_global is None and aspect_ratios_per_layer is None:
        raise ValueError("`aspect_ratios_global` and `aspect_ratios_per_layer` cannot both be None. At least one needs to be specified.")
    if aspect_ratios_per_layer:
        if len(aspect_ratios_per_layer) != n_predictor_layers:
            raise ValueError("It must be either aspect_ratios_per_layer is None or len(aspect_ratios_per_layer) == {}, but "
                             "len(aspect_ratios_per_layer) == {}.".format(n_predictor_layers, len(aspect_ratios_per_layer)))
    if (min_scale is None or max_scale is None) and scales is None:
        raise ValueError("Either `min_scale` and `max_scale` or `scales` need to be specified.")
    if scales:
        if len(scales) != n_predictor_layers + 1:
            raise ValueError("It must be either scales is None or len(scales) == {}, but len(scales) == {}.".format(
                n_predictor_layers + 1, len(scales)))
    else:
        scales = np.linspace(min_scale, max_scale, n_predictor_layers + 1)
    if len(variances) != 4:
        raise ValueError("4 variance values must be pased, but {} values were received.".format(len(variances)))
    if np.any(np.array(variances) <= 0):
        raise ValueError("All variances must be >0, but the variances given are {}".format(variances))
    if (steps is not None) and (len(steps) != n_predictor_layers):
        raise ValueError("You must provide at least one step value per predictor layer.")
    if (offsets is not None) and (len(offsets) != n_predictor_layers):
        raise ValueError("You must provide at least one offset value per predictor layer.")
    ars = aspect_ratios_per_layer if aspect_ratios_per_layer else [aspect_ratios_global] * n_predictor_layers
    n_boxes = [n_boxes_for(ar, two_boxes_for_ar1) for ar in ars]
    steps = steps if steps is not None else [None] * n_predictor_layers
    offsets = offsets if offsets is not None else [None] * n_predictor_layers
    return list(scales), ars, n_boxes, steps, offsets


class _ReluLink:
    """Training step, two ReLU convolutions in a row where the upper one is the lower one's ONLY consumer (conv2_1 -> conv2_2, conv3_1 ->
    conv3_2 -> conv3_3, conv4_1 -> conv4_2 -> conv4_3): the upper layer's data gradient can leave its kernel already masked by its
    input > 0 -- which is the lower layer's threshold_backward -- so the lower layer skips its pass over (dL/dy, y).  The link is how the
    two autograd nodes agree: the upper node's backward sets `masked` only when its kernel really applied the mask, the lower node's
    backward (which autograd runs after it) consumes the flag and falls back to its own mask otherwise."""
    __slots__ = ("masked", "partial")

    def __init__(self):
        self.masked = False
        self.partial = None      # the masked gradient's channel sums by workgroup ([rows, C] float32): the lower layer's bias-gradient partials


class _ConvBiasActFn(torch.autograd.Function):
    """A convolution layer of the TRAINING step with libssdhip's MFMA kernel in the forward pass (convolution + bias + ReLU, one
    kernel, bf16 NHWC -- the same kernels the inference path runs) and libssdhip's data / weight gradients behind it
    (`_conv_input_weight_grads`: since round 6 every layer of SSD300 / SSD512 except a 4 x 4 or grouped convolution; the framework's
    convolution_backward is the fallback for what the kernels do not cover).  `run` is the libssdhip thunk picked for this layer
    shape: (x_bf16, w_bf16, b_bf16) -> y."""

    @staticmethod
    def forward(ctx, x, weight, bias, run, stride, padding, dilation, relu, wb=None, bb=None, wt=None, link_in=None, link_out=None):
        xb = x.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if wb is None:                                   # no bf16 shadow of the parameters at hand: cast here
            wb = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            bb = bias.detach().to(torch.bfloat16) if bias is not None else None
        y = run(xb, wb, bb)
        # (wt: the data gradient's filters, built with the shadows -- None: built in backward; saved like the others so that autograd's
        #  version check covers it when the shadows are refreshed between this forward and its backward)
        ctx.save_for_backward(xb, wb, y if relu else None, wt)
        ctx.conf = (stride, padding, dilation, relu, weight.dtype, None if bias is None else bias.dtype, x.dtype)
        # link_in: x is the ReLU output of a layer that feeds nothing else (_ReluLink); link_out: the same towards this layer's consumer
        ctx.links = (link_in, link_out if relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        xb, wb, y, wt = ctx.saved_tensors
        stride, padding, dilation, relu, wdt, bdt, xdt = ctx.conf
        gy = gy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        want_gb = bdt is not None and ctx.needs_input_grad[2]
        partial = None
        import os
        if (relu and not ctx.needs_input_grad[0] and gy.is_cuda and tuple(wb.shape) == (64, 3, 3, 3) and stride == (1, 1)
                and padding == (1, 1) and dilation == (1, 1) and os.environ.get("SSDHIP_NO_CONV1_1_BWD", "0") != "1"):
            # the first layer: no data gradient, so the masked gradient is only ever summed -- ReLU mask, bias gradient and weight gradient
            # in ONE pass that writes nothing but partial sums (csrc/ssdhip_train.hip, conv1_1_bwd_kernel)
            gw, gb = nat.conv1_1_backward(gy, y, xb)
            return None, gw.to(wdt), (gb.to(bdt) if want_gb else None), None, None, None, None, None, None, None, None, None, None
        link_in, link_out = ctx.links
        if relu:
            premasked = link_out is not None and link_out.masked
            if link_out is not None:
                partial, link_out.partial = link_out.partial, None
                link_out.masked = False                  # consumed: the consumer's next backward sets it again
            fused = None
            if premasked:
                # dL/dy arrived masked by y > 0 from the consumer's data-gradient kernel (_ReluLink), its channel sums beside it
                if want_gb and partial is None:
                    partial = nat.channel_sums_partial(gy)
            else:
                # ReLU mask and the per-workgroup channel sums of the bias gradient in ONE libssdhip pass (csrc/ssdhip_train.hip); the rows
                # are added by the weight gradient's reduction launch where that is ours, by one framework reduction otherwise
                fused = nat.relu_bwd_bias(gy, y, reduce=False)
                if fused is not None:
                    gy, partial = fused
                else:
                    gy = torch.ops.aten.threshold_backward(gy, y, 0)
        gx, gw, gb = _conv_input_weight_grads(gy, xb, wb, stride, padding, dilation, ctx.needs_input_grad[0], wt,
                                              partial if want_gb else None, link_in)
        if want_gb:
            if gb is None:
                gb = nat.row_sums(partial) if partial is not None else gy.sum(dim=(0, 2, 3), dtype=torch.float32)
            gb = gb.to(bdt)
        else:
            gb = None
        return (gx.to(xdt) if gx is not None else None), gw.to(wdt), gb, None, None, None, None, None, None, None, None, None, None


def _conv_input_weight_grads(gy, xb, wb, stride, padding, dilation, need_x, wt=None, bias_partial=None, link_in=None):
    """dL/dx, dL/dw [and dL/db] of a convolution from the (masked) dL/dy.  Data gradient: a stride-1 'same' layer through the forward's
    MFMA kernel on the transposed, tap-flipped filters; a strided or 'valid' 3 x 3 layer the same way behind an embedding launch (round
    6).  Weight gradient: the position-grid kernel (3 x 3 'same', incl. fc6's dilation 6), the pixel GEMM (1 x 1), the tap-gathered pixel
    GEMM (any other 3 x 3) -- csrc/ssdhip_wgrad.hip.  What none of them covers goes to aten.convolution_backward (MIOpen).
    bias_partial: per-workgroup channel sums of gy ([rows, Cout] float32); the third result is their ordered sum when the weight
    gradient's reduction launch could add them on the side, None otherwise (the caller reduces them itself).
    link_in (_ReluLink): xb is the ReLU output of a layer that feeds nothing else -- where the slab kernel runs the data gradient it
    writes dL/dx masked by xb > 0 and sets the link (the layer below then skips its own mask pass)."""
    gx = None
    k = wb.shape[2]
    same = (stride == (1, 1) and k % 2 == 1 and padding == (dilation[0] * (k // 2),) * 2 and dilation[0] == dilation[1]
            and wb.shape[0] % 64 == 0 and wb.shape[1] % 64 == 0 and k in (1, 3))
    import os
    own_taps = os.environ.get("SSDHIP_NO_TAPS_BWD", "0") != "1"
    # (round 6) a strided or 'valid' 3 x 3 layer (conv6_2 / conv7_2: stride 2 behind ZeroPadding2D; conv8_2 / conv9_2: no padding):
    # dX[r] = sum_k dY[(r + pad - k) / s] w[k] is the 3 x 3 'same' convolution of Z -- zeros with dY at (1 - pad + s i) -- with the same
    # transposed, tap-flipped filters: one embedding launch (csrc/ssdhip_train.hip), then the branch below as for a 'same' layer
    embedded = (need_x and not same and own_taps and k == 3 and dilation == (1, 1) and stride[0] == stride[1] and padding[0] == padding[1]
                and padding[0] in (0, 1) and wb.shape[0] % 64 == 0 and wb.shape[1] % 64 == 0 and gy.is_cuda
                and os.environ.get("SSDHIP_NO_OWN_DGRAD", "0") != "1")
    if embedded:
        gy_full, gy = gy, nat.embed_strided(gy, xb.shape[2], xb.shape[3], stride[0], 1 - padding[0])
    if need_x and (same or embedded) and os.environ.get("SSDHIP_NO_OWN_DGRAD", "0") != "1":
        # the data gradient of a stride-1 'same' convolution IS a 'same' convolution of dL/dy with the filters transposed
        # (Cin <-> Cout) and their taps flipped: the forward's MFMA kernel runs it, no bias, no activation
        if wt is None:                                   # (the shadow set hands the transposed filters over: csrc/ssdhip_optim.hip)
            wt = wb.flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
        # (the deep 3x3 layers through the slab kernel, csrc/ssdhip_convh.hip: bit-identical and faster, r02o)
        halo = (k == 3 and dilation[0] == 1 and wt.shape[0] % 128 == 0 and wt.shape[1] % 128 == 0
                and os.environ.get("SSDHIP_NO_HALO", "0") != "1")
        # (small maps -- conv5_x, fc6 with its dilation -- through the image-resident kernel, csrc/ssdhip_convimg.hip, where it fills the chip)
        image = (k == 3 and gy.shape[2] * gy.shape[3] <= 384 and gy.shape[0] * (wt.shape[0] // 64) >= 128
                 and nat.conv3x3_image_supported(gy, wt, dilation[0]) and os.environ.get("SSDHIP_NO_IMAGE", "0") != "1")
        # (a 64-channel dL/dy -- conv1_2 -- through the resident-filter kernel of the Cin = 64 layers, csrc/ssdhip_conv64.hip: the same bits as
        #  the implicit-GEMM kernel in half its time, 430 -> 215 us at 300 x 300 / batch 32)
        c64 = (k == 3 and dilation[0] == 1 and wt.shape[1] == 64 and wt.shape[0] % 64 == 0 and os.environ.get("SSDHIP_NO_C64_DGRAD", "0") != "1")
        # (round 6: the 1 x 1 layers on small maps -- fc7, conv6_1 -- through the same kernel's one-step-per-slice form)
        image1 = (k == 1 and gy.shape[2] * gy.shape[3] <= 384 and gy.shape[0] * (wt.shape[0] // 64) >= 128
                  and nat.conv2d_image_supported(gy, wt) and os.environ.get("SSDHIP_IMAGE2", "1") != "0"
                  and os.environ.get("SSDHIP_NO_IMAGE", "0") != "1")
        masked = None
        if (link_in is not None and halo and same and not image and gy.dtype == torch.bfloat16 and xb.dtype == torch.bfloat16
                and os.environ.get("SSDHIP_NO_MASKED_DGRAD", "0") != "1"):
            masked = nat.conv3x3_halo_masked(gy, wt, xb, sums=os.environ.get("SSDHIP_NO_MASKED_SUMS", "0") != "1")
        if masked is not None:
            gx, link_in.partial = masked if isinstance(masked, tuple) else (masked, None)
            link_in.masked = True
        elif image:
            gx = nat.conv3x3_image(gy, wt, None, dilation=dilation[0], relu=False)
        elif image1:
            gx = nat.conv2d_image(gy, wt, None, relu=False)
        elif c64:
            gx = nat.conv3x3_c64(gy, wt, None, relu=False, pool=False)
        else:
            gx = nat.conv2d_same(gy, wt, None, dilation=dilation[0], relu=False, variant=7 if halo else None)
    if embedded:
        gy = gy_full
    gw, gb = None, None
    if (k == 3 and stride == (1, 1) and padding == (1, 1) and dilation == (1, 1) and os.environ.get("SSDHIP_NO_OWN_WGRAD", "0") != "1"):
        # the weight gradient through libssdhip's MFMA kernel (csrc/ssdhip_wgrad.hip; float32, fixed summation order); None: geometry
        # not covered (3 input channels, predictor heads whose channel counts are not multiples of 64)
        if xb.shape[1] % 64 == 0 and gy.shape[1] % 64 == 0:
            got = nat.conv3x3_wgrad(xb, gy, bias_partial=bias_partial)
            if got is not None and bias_partial is not None:
                gw, gb = got
            else:
                gw = got
    if (gw is None and k == 1 and stride == (1, 1) and padding == (0, 0) and gy.is_cuda and xb.shape[1] % 128 == 0 and gy.shape[1] % 128 == 0
            and os.environ.get("SSDHIP_NO_OWN_WGRAD", "0") != "1"):
        # the 1 x 1 layers (fc7, conv6_1 ... conv9_1): the weight gradient is a GEMM over the pixels (csrc/ssdhip_wgrad.hip,
        # conv1x1_wgrad_kernel; float32, fixed summation order), the bias partials ride in its reduction launch
        got = nat.conv1x1_wgrad(xb, gy, bias_partial=bias_partial)
        if got is not None and bias_partial is not None:
            gw, gb = got
        else:
            gw = got
    if (gw is None and k == 3 and own_taps and gy.is_cuda and stride[0] == stride[1] and padding[0] == padding[1] and dilation[0] == dilation[1]
            and xb.shape[1] % 128 == 0 and gy.shape[1] % 128 == 0 and os.environ.get("SSDHIP_NO_OWN_WGRAD", "0") != "1"):
        # (round 6) the other 3 x 3 layers -- fc6's dilation, the strided and the 'valid' extras -- through the tap-gathered pixel GEMM
        # (csrc/ssdhip_wgrad.hip, conv_taps_wgrad_kernel): with it the training step holds no framework convolution
        got = nat.conv3x3_taps_wgrad(xb, gy, stride[0], padding[0], dilation[0], bias_partial=bias_partial)
        if got is not None and bias_partial is not None:
            gw, gb = got
        else:
            gw = got
    masks = [need_x and gx is None, gw is None, False]
    if masks[0] or masks[1]:
        gx_m, gw_m, _ = torch.ops.aten.convolution_backward(gy, xb, wb, None, list(stride), list(padding), list(dilation), False, [0, 0],
                                                            1, masks)
        if gx is None and need_x:
            gx = gx_m
        if gw is None:
            gw = gw_m
    return gx, gw, gb


class _ConvBiasActPoolFn(torch.autograd.Function):
    """Conv2D(relu) -> MaxPooling2D(2, 2, 'same') of the TRAINING step (pool1 .. pool3) as one autograd node: forward = the layer's
    MFMA kernel + the one-pass pooling kernel (conv1_2 -> pool1: ONE launch that writes both maps, round 6); backward = max-pool gradient, ReLU mask and bias gradient in ONE pass over the
    full-resolution map (csrc/ssdhip_train.hip, maxpool2_relu_bwd_bias_kernel) -- the unmasked full-resolution gradient is never
    written -- then the convolution's gradients as in _ConvBiasActFn."""

    @staticmethod
    def forward(ctx, x, weight, bias, run, stride, padding, dilation, wb=None, bb=None, wt=None, link_in=None):
        xb = x.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if wb is None:
            wb = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            bb = bias.detach().to(torch.bfloat16) if bias is not None else None
        ctx.link_in = link_in
        import os
        if (xb.shape[1] == 64 and wb.shape[2:] == (3, 3) and stride == (1, 1) and padding == (1, 1) and dilation == (1, 1)
                and wb.shape[0] % 64 == 0 and xb.is_cuda and os.environ.get("SSDHIP_NO_POOL_KEEP", "0") != "1"):
            # round 6: conv1_2 -> pool1 as ONE launch that writes the activation AND the pooled map (csrc/ssdhip_conv64.hip, KEEP): the
            # pooling pass read the 368 MB map back (~95 us of the step)
            y, p = nat.conv3x3_c64_pool_keep(xb, wb, bb, relu=True)
        else:
            # fourth session: conv2_2 -> pool2 and conv3_3 -> pool3 the same way on the slab kernel (csrc/ssdhip_convh.hip, KEEP)
            kept = None
            if (xb.shape[1] % 128 == 0 and wb.shape[0] % 128 == 0 and wb.shape[2:] == (3, 3) and stride == (1, 1) and padding == (1, 1)
                    and dilation == (1, 1) and xb.is_cuda and os.environ.get("SSDHIP_NO_POOL_KEEP", "0") != "1"
                    and os.environ.get("SSDHIP_NO_HALO", "0") != "1" and os.environ.get("SSDHIP_NO_HALO_POOL_KEEP", "0") != "1"):
                kept = nat.conv3x3_halo_pool_keep(xb, wb, bb, relu=True)
            if kept is not None:
                y, p = kept
            else:
                y = run(xb, wb, bb)
                p = nat.bias_act_maxpool(y, None, 2, 2, 0, True, relu=False)
        ctx.save_for_backward(xb, wb, y, wt)
        ctx.conf = (stride, padding, dilation, weight.dtype, None if bias is None else bias.dtype, x.dtype)
        return p

    @staticmethod
    def backward(ctx, gp):
        xb, wb, y, wt = ctx.saved_tensors
        stride, padding, dilation, wdt, bdt, xdt = ctx.conf
        gp = gp.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        fused = nat.maxpool2_relu_bwd_bias(y, gp, reduce=False)
        if fused is None:
            raise RuntimeError("channel count not supported by the fused pooling backward (the forward checks it)")
        gy, partial = fused
        want_gb = bdt is not None and ctx.needs_input_grad[2]
        gx, gw, gb = _conv_input_weight_grads(gy, xb, wb, stride, padding, dilation, ctx.needs_input_grad[0], wt,
                                              partial if want_gb else None, ctx.link_in)
        if want_gb:
            gb = (gb if gb is not None else nat.row_sums(partial)).to(bdt)
        else:
            gb = None
        return (gx.to(xdt) if gx is not None else None), gw.to(wdt), gb, None, None, None, None, None, None, None, None


class _PackedHeadFn(torch.autograd.Function):
    """The two predictor heads of one source map in the TRAINING step as one libssdhip node (round 4): conf and loc filters packed along
    Cout (zero rows up to a multiple of 128), forward = the slab kernel (no activation), data gradient = the slab kernel on the flipped /
    transposed pack, weight gradient = ssdhip_conv3x3_wgrad on the packed gradient, bias gradient = one reduction -- instead of two
    framework convolutions forward and four backward per map (MIOpen: 1.7 ms of a 12.7 ms step, profiles/r04za).  Returns the packed
    (B, Cp, H, W) bf16 map; the caller slices conf / loc out of it (reference: models/keras_ssd300.py:322-335)."""

    @staticmethod
    def forward(ctx, x, wc, bc, wl, bl, pw, pb, pwt):
        """pw / pb / pwt: the packed bf16 filters [conf | loc | zero rows], biases and transposed / flipped filters of this source map,
        kept up to date with the parameters by the model's shadow set (SSDModel._packed_head_shadow): nothing is concatenated here."""
        xb = x.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = nat.conv3x3_halo(xb, pw, pb, relu=False, pool=False)
        ctx.save_for_backward(xb, pw, pwt)
        ctx.conf = (wc.shape[0], wl.shape[0], wc.dtype, bc.dtype, x.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        xb, w, wt = ctx.saved_tensors
        nc, nl, wdt, bdt, xdt = ctx.conf
        gyb = gy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = nat.conv2d_same(gyb, wt, None, dilation=1, relu=False, variant=7).to(xdt)
        partial = nat.channel_sums_partial(gyb)              # per-workgroup channel sums; the weight gradient's reduction launch adds them
        got = nat.conv3x3_wgrad(xb, gyb, bias_partial=partial)
        if got is None:
            raise RuntimeError("packed predictor head: weight-gradient geometry not covered")
        gw, gb = got if partial is not None else (got, gyb.float().sum(dim=(0, 2, 3)))
        return (gx, gw[:nc].to(wdt), gb[:nc].to(bdt), gw[nc:nc + nl].to(wdt), gb[nc:nc + nl].to(bdt), None, None, None)


class _AssembleTrainFn(torch.autograd.Function):
    """Reshape + Concatenate + softmax + AnchorBoxes + Concatenate of the TRAINING step (models/keras_ssd300.py:363-419) as one autograd
    node over the packed head maps: forward = ssdhip_assemble_predictions_strided_bf16 (the inference path's one-launch assembly),
    backward = ssdhip_assemble_predictions_backward_bf16 (softmax backward and the scatter into the packed layout, one launch) --
    instead of six slices, three concatenations, a softmax and an index_select forward and their ~15 kernels backward."""

    @staticmethod
    def forward(ctx, anchors, n_classes, n_boxes, *ys):
        pred = nat.assemble_predictions([y.detach() for y in ys], [None] * len(ys), [None] * len(ys), [None] * len(ys), list(n_boxes),
                                        anchors, n_classes)
        ctx.save_for_backward(pred)
        ctx.conf = (n_classes, tuple(n_boxes), tuple(tuple(y.shape) for y in ys))
        return pred

    @staticmethod
    def backward(ctx, g):
        (pred,) = ctx.saved_tensors
        n_classes, n_boxes, shapes = ctx.conf
        grads = nat.assemble_predictions_backward(g.float(), pred, shapes, n_boxes, n_classes)
        return (None, None, None) + tuple(grads)


class _MaxPoolFn(torch.autograd.Function):
    """max_pool2d of a bf16 NHWC map in the training step: libssdhip forward (one pass) and backward (gather, deterministic)."""

    @staticmethod
    def forward(ctx, x, kernel, stride, pad, ceil_mode):
        y = nat.bias_act_maxpool(x, None, kernel, stride, pad, ceil_mode, relu=False)
        ctx.save_for_backward(x)
        ctx.conf = (kernel, stride, pad)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        kernel, stride, pad = ctx.conf
        return nat.maxpool_bwd(x, gy.to(torch.bfloat16), kernel, stride, pad), None, None, None, None


class GraphedInference:
    """`model(images)` (forward + DecodeDetections) of a fixed input shape as a HIP graph.

    The graph reads the tensor handed to the constructor (`static_in`) and writes `static_out`: calling the object with another
    tensor copies it into `static_in` first (one device copy); the returned tensor is overwritten by the next call.  Capture happens
    after `warmup` eager steps on the capture stream, so the per-shape kernel autotune, the workspaces and the side stream of the
    predictor heads exist before anything is recorded (allocations, host -> device copies and timing syncs are illegal inside a
    capture).  Inference only (no_grad)."""

    def __init__(self, model, images, warmup=3, fn=None):
        if not images.is_cuda:
            raise ValueError("HIP graphs need a CUDA/HIP tensor")
        run = fn if fn is not None else model            # fn: another callable of the model on the same input (model.head_outputs)
        self.model = model
        self.static_in = images
        dev = images.device
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        had = model.__dict__.get("_head_overlap")
        model.__dict__["_head_overlap"] = os.environ.get("SSDHIP_GRAPH_HEAD_OVERLAP", "4")   # two streams inside the graph: explicit dependencies, no allocator subtleties
        try:
            with torch.cuda.stream(self.stream), torch.no_grad():
                for _ in range(max(1, warmup)):
                    run(self.static_in)
            torch.cuda.current_stream(dev).wait_stream(self.stream)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="relaxed"):
                self.static_out = run(self.static_in)
        finally:
            model.__dict__["_head_overlap"] = had
        # (see __call__: an OLDER graph of a model must not be replayed on a foreign stream once a newer one exists)
        self._epoch = model.__dict__.get("_graph_epoch", 0) + 1
        model.__dict__["_graph_epoch"] = self._epoch

        # What the recorded kernels read besides `static_in`: the parameters' own storage (in-place updates -- optimizer steps,
        # load_state_dict, load_keras_weights -- are seen by the next replay) and the PACKED head filters, separate tensors built from
        # the conf / loc weights, and the other tensors DERIVED from parameters (the fragment-packed conv7_1 ... conv9_2 filters of
        # SSD300's one-launch tail, the float32 copy of L2Normalization's gamma in a bf16 model): all of them are refreshed in place
        # when the parameter they come from changed (`_refresh_derived_weights`, keyed on `_derived_weights_key`).  A parameter
        # whose storage was REPLACED (`conv.weight = nn.Parameter(...)`, `param.data = t`) is something the graph cannot follow.
        self._param_ptrs = tuple(p.data_ptr() for p in model.parameters())
        self._derived_key = model._derived_weights_key()

    def __call__(self, images=None):
        if images is not None and images.data_ptr() != self.static_in.data_ptr():
            if tuple(images.shape) != tuple(self.static_in.shape) or images.dtype != self.static_in.dtype:
                raise ValueError("this graph was captured for images of shape %s / %s, got %s / %s" % (
                    tuple(self.static_in.shape), self.static_in.dtype, tuple(images.shape), images.dtype))
            self.static_in.copy_(images, non_blocking=True)
        if tuple(p.data_ptr() for p in self.model.parameters()) != self._param_ptrs:
            raise RuntimeError("a parameter's storage was replaced after the graph was captured: call model.graphed(...) again")
        key = self.model._derived_weights_key()
        if key != self._derived_key:
            self.model._refresh_derived_weights()
            self._derived_key = key
        cur = torch.cuda.current_stream(self.static_in.device)
        if self.model.__dict__.get("_graph_epoch", 0) != self._epoch and cur.cuda_stream != self.stream.cuda_stream:
            # Another graph of this model was captured after this one.  On ROCm 7.2 replaying the OLDER of two such graphs on a stream
            # other than its capture stream segfaults inside hipGraphLaunch (tools/debug_two_graphs.py: 2 graphs + foreign stream
            # crashes, 1 graph or the capture stream does not; profiles/r06zq_two_steps_in_flight_negative.txt) -- so it is replayed
            # on its capture stream, ordered behind and in front of the caller's stream.
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.graph.replay()
            cur.wait_stream(self.stream)
            return self.static_out
        self.graph.replay()
        return self.static_out


class SSDModel(nn.Module):
    """Base class: subclasses define `features(x) -> list of predictor feature maps` plus
    `conf_heads`, `loc_heads` (ModuleLists) and `priorboxes` (ModuleList of AnchorBoxes)."""

    def __init__(self, image_size, n_classes, mode, l2_regularization, subtract_mean, divide_by_stddev, swap_channels,
                 confidence_thresh, iou_threshold, top_k, nms_max_output_size, coords, normalize_coords):
        super().__init__()
        if mode not in ('training', 'inference', 'inference_fast'):
            raise ValueError("`mode` must be one of 'training', 'inference' or 'inference_fast', but received '{}'.".format(mode))
        self.img_height, self.img_width, self.img_channels = image_size
        self.n_classes = n_classes + 1               # incl. background, as in the reference (:175)
        self.mode = mode
        self.l2_regularization = l2_regularization
        self.subtract_mean = subtract_mean
        self.divide_by_stddev = divide_by_stddev
        self.swap_channels = swap_channels
        self.fused_inference = True         # bf16 + no_grad on a GPU: graph glue runs in libssdhip (csrc/ssdhip_layers.hip)
        self.fused_training = True          # grad mode on a GPU in bf16 (autocast): convolution forwards run in libssdhip too
        self.decoder = None
        if mode != 'training':
            layer = DecodeDetections if mode == 'inference' else DecodeDetectionsFast
            self.decoder = layer(confidence_thresh=confidence_thresh, iou_threshold=iou_threshold, top_k=top_k,
                                 nms_max_output_size=nms_max_output_size, coords=coords,
                                 normalize_coords=normalize_coords, img_height=self.img_height,
                                 img_width=self.img_width, name='decoded_predictions')
        self._anchor_cache = {}
        self._packed_heads = {}

    # -- fused inference path: every conv is followed by ONE libssdhip pass (bias + ReLU [+ max-pool]) instead of the
    #    2-3 elementwise kernels PyTorch launches; bit-identical results (see csrc/ssdhip_layers.hip) ----------------
    def _fused(self, x, conv=None):
        return (self.fused_inference and x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled()
                and (conv is None or (conv.weight.dtype == torch.bfloat16 and conv.out_channels % 8 == 0)))

    @staticmethod
    def _conv_nobias(conv, x):
        return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)

    # Which kernel runs a convolution of the fused path: 'miopen' (F.conv2d + one libssdhip bias/ReLU[/pool] pass) or
    # 'igemm' (libssdhip's implicit-GEMM MFMA kernel with the bias/ReLU epilogue, csrc/ssdhip_conv.hip).  SSDHIP_CONV =
    # auto (default: time libssdhip's variants once per layer shape, keep the fastest), auto_miopen (MIOpen competes too),
    # igemm, miopen.
    _conv_choice = {}

    @staticmethod
    def _igemm_ok(conv, x):
        k = conv.kernel_size[0]
        return (k in (1, 3) and conv.kernel_size[1] == k and conv.stride == (1, 1) and conv.groups == 1
                and conv.dilation[0] == conv.dilation[1] and conv.padding == (conv.dilation[0] * (k // 2),) * 2
                and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0 and conv.bias is not None)

    @staticmethod
    def _igemm_general_ok(conv, x):
        """Strided / partially padded 3x3 and 1x1 layers (conv6_2 ... conv9_2) for nat.conv2d."""
        k = conv.kernel_size[0]
        return (k in (1, 3) and conv.kernel_size[1] == k and conv.stride[0] == conv.stride[1] and 1 <= conv.stride[0] <= 4
                and conv.groups == 1 and conv.dilation[0] == conv.dilation[1] and isinstance(conv.padding, tuple)
                and conv.padding[0] == conv.padding[1] and 0 <= conv.padding[0] <= conv.dilation[0] * (k // 2)
                and conv.padding_mode == 'zeros' and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0
                and conv.bias is not None)

    @staticmethod
    def _few_tiles(m_pixels, cout):
        """Fewer 128 x 128 output tiles than a third of the CUs: the layer's one-pass kernels leave most of the chip idle while a few
        workgroups walk their whole K loop (the SSD extra layers behind fc7)."""
        return -(-m_pixels // 128) * -(-cout // 128) <= 100

    @staticmethod
    def _splitk_measured_regime(x):
        """Where the split-K form was measured inside the graphed step (profiles/r03p_*: the SSD300 / SSD512 extra layers behind fc7 at
        batch 32: maps of at most 19 x 19 ... 32 x 32 pixels, a batch that fills the K ranges): only there is it taken without a timing
        run.  Everywhere else (small batches, where `_few_tiles` also covers conv3_x ... fc7) it is an ordinary autotune candidate."""
        return x.shape[0] >= 16 and x.shape[2] * x.shape[3] <= 32 * 32

    @staticmethod
    def _halo_ok(conv, x):
        """csrc/ssdhip_convh.hip: 3x3, dilation 1, Cin and Cout multiples of 128 (maps up to 94 wide on the padded position grid,
        wider ones and the pooled form on 2-D tiles)."""
        import os
        return (conv.kernel_size == (3, 3) and conv.dilation == (1, 1) and conv.in_channels % 128 == 0
                and conv.out_channels % 128 == 0 and os.environ.get("SSDHIP_NO_HALO", "0") != "1")

    @staticmethod
    def _image_ok(conv, x):
        """csrc/ssdhip_convimg.hip: 3x3 'same' with any dilation, the whole map of an image (at most 384 pixels) resident in LDS, one
        tile per (image, 128 output channels) -- offered where that gives at least half a chip's worth of tiles."""
        import os
        return (conv.kernel_size == (3, 3) and x.shape[2] * x.shape[3] <= 384 and conv.in_channels % 64 == 0
                and conv.out_channels % 64 == 0 and 1 <= conv.dilation[0] <= 16
                and x.shape[0] * (conv.out_channels // 64) >= 128 and os.environ.get("SSDHIP_NO_IMAGE", "0") != "1")

    @staticmethod
    def _image2_ok(conv, x):
        """Round 6, csrc/ssdhip_convimg.hip's general form: 1 x 1 layers (fc7, conv6_1: one step per 64-channel slice of the resident
        image) and strided / partially padded 3 x 3 layers (conv6_2) on maps of at most 384 pixels, where one image x 64 output channels
        per tile gives at least half a chip's worth of tiles."""
        import os
        k = conv.kernel_size[0]
        if (os.environ.get("SSDHIP_IMAGE2", "1") == "0" or os.environ.get("SSDHIP_NO_IMAGE", "0") == "1" or conv.kernel_size[1] != k
                or conv.stride[0] != conv.stride[1] or conv.dilation[0] != conv.dilation[1] or not isinstance(conv.padding, tuple)
                or conv.padding[0] != conv.padding[1] or conv.groups != 1 or conv.padding_mode != 'zeros'):
            return False
        return (x.shape[0] * (conv.out_channels // 64) >= 128
                and nat.conv2d_image_supported(x, conv.weight, conv.stride[0], conv.padding[0], conv.dilation[0]))

    def _pick(self, key, candidates):
        """candidates: {name: thunk}; returns the name of the fastest (timed once per key with events)."""
        import os
        mode = os.environ.get("SSDHIP_CONV", "auto")
        if mode in candidates:
            return mode
        prefer = os.environ.get("SSDHIP_PREFER")              # A/B aid: this candidate wherever it is offered, the autotune elsewhere
        if prefer and prefer in candidates:
            return prefer
        if mode == "igemm" and "igemm" not in candidates:
            return "miopen"
        hit = SSDModel._conv_choice.get(key)
        if hit is None:
            if mode != "auto_miopen" and len(candidates) > 1 and "miopen" in candidates:
                # Measured on MI355X (profiles/r01*, r02b_bench.json): libssdhip's kernels beat MIOpen + one bias/ReLU pass on every
                # SSD300 / SSD512 layer, and on shapes where MIOpen has no tuned solver its find step falls back to a naive kernel
                # (~20 s of probing per process).  MIOpen is only timed when asked for (SSDHIP_CONV=auto_miopen) or when it is the
                # only candidate.
                candidates = {k: v for k, v in candidates.items() if k != "miopen"}
                if len(candidates) == 1:
                    hit = next(iter(candidates))
                    SSDModel._conv_choice[key] = hit
                    return hit
            best, hit, times = None, None, {}
            # the library candidate goes last: when MIOpen falls back to its naive solver for a shape (tens of ms per call) the
            # single-call probe below drops it without paying for the bursts
            for name in sorted(candidates, key=lambda n: n == "miopen"):
                fn = candidates[name]
                fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if best is not None:
                    a.record()
                    fn()
                    b.record()
                    b.synchronize()
                    if a.elapsed_time(b) > 3.0 * best / 4.0:
                        continue
                t = None
                for _ in range(3):                       # best of three bursts of six: one noisy burst does not decide a layer
                    a.record()
                    for _ in range(6):
                        fn()
                    b.record()
                    b.synchronize()
                    t = a.elapsed_time(b) if t is None else min(t, a.elapsed_time(b))
                times[name] = t
                if best is None or t < best:
                    best, hit = t, name
            # Near-ties go to the deeper-pipelined form: a back-to-back burst is L2-warm and host-paced, and on fc6 / conv6_1 it has put
            # the two-stage kernel a few per cent ahead of the three-stage one that is 10-25 % faster inside the step
            # (profiles/r04s_step_timeline.json against r04n: fc6 142 vs 128 us, conv6_1 28 vs 22 us).
            # (plain convolutions only: among the pooled forms "halo" is the UNFUSED slab kernel + a pooling pass.)
            # Only the two-stage kernel (or the library) loses a near-tie: a measured winner among the deeper forms stays the winner,
            # and among the deeper forms within 8 % of it the FASTEST takes over, not the first of a fixed list.
            if key and key[0] == "act" and hit in ("igemm", "miopen"):
                near = [(times[name], name) for name in ("halo", "image", "igemm6", "igemm5")  # fc6 in the step: igemm6 128 us, igemm 142, igemm5 ~165 (r04n / r04s / r04zz)
                        if name in times and times[name] <= 1.08 * best]
                if near:
                    hit = min(near)[1]
            SSDModel._conv_choice[key] = hit
        return hit

    @staticmethod
    def _gemm_1x1(conv, x, relu):
        b, c, h, w = x.shape
        x2 = x.permute(0, 2, 3, 1).reshape(b * h * w, c)                 # a view of channels_last memory
        w2 = conv.weight.reshape(conv.out_channels, c)
        y2 = (torch._addmm_activation(conv.bias, x2, w2.t(), use_gelu=False) if relu else torch.addmm(conv.bias, x2, w2.t()))
        return y2.view(b, h, w, conv.out_channels).permute(0, 3, 1, 2)

    def conv_act(self, conv, x, relu=True, link_in=None, link_out=None):
        """Conv2D(activation='relu' | None).  link_in / link_out (_ReluLink, training step only; see `relu_link`): x is the ReLU output of a
        layer that feeds nothing but this one / this layer's output feeds exactly one layer, which holds the same link as its link_in."""
        if self._fused(x, conv):
            import os
            k = conv.kernel_size[0]
            if (conv.in_channels == 3 and conv.out_channels == 64 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
                    and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.bias is not None):
                return nat.conv3x3_cin3(x, conv.weight, conv.bias, relu=relu)
            cands = {"miopen": lambda: nat.bias_act(self._conv_nobias(conv, x), conv.bias, relu=relu)}
            if self._igemm_ok(conv, x):
                cands["igemm"] = lambda: nat.conv2d_same(x, conv.weight, conv.bias, dilation=conv.dilation[0], relu=relu)
                # the three-stage / 32-channel-slice / 3-workgroups-per-CU variant wins on the shallow-K layers (Cin = 64)
                cands["igemm6"] = lambda: nat.conv2d_same(x, conv.weight, conv.bias, dilation=conv.dilation[0], relu=relu, variant=6)
                if x.shape[0] * x.shape[2] * x.shape[3] <= 128 * 128:          # at most one workgroup per CU: the deepest ring too
                    cands["igemm5"] = lambda: nat.conv2d_same(x, conv.weight, conv.bias, dilation=conv.dilation[0], relu=relu, variant=5)
                if self._few_tiles(x.shape[0] * x.shape[2] * x.shape[3], conv.out_channels):
                    # the split-K form: the K ranges of a tile side by side on otherwise idle CUs
                    cands["splitk"] = lambda: nat.conv2d(x, conv.weight, conv.bias, stride=1, padding=conv.padding[0],
                                                         dilation=conv.dilation[0], relu=relu, variant=8)
                if conv.in_channels == 64 and k == 3 and conv.dilation[0] == 1:
                    cands["c64"] = lambda: nat.conv3x3_c64(x, conv.weight, conv.bias, relu=relu, pool=False)
                if self._halo_ok(conv, x):
                    cands["halo"] = lambda: nat.conv2d_same(x, conv.weight, conv.bias, dilation=1, relu=relu, variant=7)
                if self._image_ok(conv, x):
                    # one image per tile, the dilated taps as per-lane LDS addresses (csrc/ssdhip_convimg.hip): fc6
                    cands["image"] = lambda: nat.conv3x3_image(x, conv.weight, conv.bias, dilation=conv.dilation[0], relu=relu)
                if k == 1 and self._image2_ok(conv, x):
                    # round 6: a 1 x 1 layer with the image's 64-channel slices resident in LDS, one step per slice (fc7 41 -> ~17 us,
                    # conv6_1 23 -> ~10 us: the implicit-GEMM tiles move 2.5 x the bytes per FLOP from L2).  Taken without a timing
                    # run, like the split-K form below: a back-to-back burst of these kernels is L2-warm and host-paced.
                    cands["image"] = lambda: nat.conv2d_image(x, conv.weight, conv.bias, relu=relu)
                    if (os.environ.get("SSDHIP_CONV", "auto") in ("auto", "auto_miopen") and not os.environ.get("SSDHIP_PREFER")
                            and x.shape[0] >= 16):
                        return cands["image"]()
                if k == 1 and os.environ.get("SSDHIP_GEMM_1X1", "0") == "1":
                    # a 1 x 1 layer on NHWC memory IS a plain GEMM ([B H W, Cin] x [Cin, Cout] + bias, ReLU).  Opt-in: the library's
                    # (hipBLASLt through torch, bias / activation in its epilogue) was measured on fc7 and conv6_1 inside the step and is
                    # no faster than the implicit-GEMM kernels (profiles/r05zd_library_gemm_for_1x1_layers_ab.txt: 2.119 / 2.112 / 2.119 ms)
                    cands["gemm"] = lambda: self._gemm_1x1(conv, x, relu)
            elif self._igemm_general_ok(conv, x):
                # the extra layers: small maps, one workgroup per CU at most -- the deeper LDS rings (loads three / two steps ahead)
                # hide the L2 latency that the two-stage kernel exposes on every K-step
                ho = (x.shape[2] + 2 * conv.padding[0] - conv.dilation[0] * (k - 1) - 1) // conv.stride[0] + 1
                wo = (x.shape[3] + 2 * conv.padding[0] - conv.dilation[0] * (k - 1) - 1) // conv.stride[0] + 1
                names = (("igemm", None), ("igemm5", 5), ("igemm6", 6))
                if self._few_tiles(x.shape[0] * ho * wo, conv.out_channels):
                    names += (("splitk", 8),)
                for nm, v in names:
                    cands[nm] = lambda v=v: nat.conv2d(x, conv.weight, conv.bias, stride=conv.stride[0], padding=conv.padding[0],
                                                       dilation=conv.dilation[0], relu=relu, variant=v)
                if (self._halo_ok(conv, x) and conv.stride[0] in (1, 2) and conv.padding[0] in (0, 1) and x.shape[3] <= 94
                        and x.shape[2] + 2 * conv.padding[0] >= 3 and x.shape[3] + 2 * conv.padding[0] >= 3):
                    # the slab kernel keeps the strided / cropped positions of the stride-1 'same' result: redundant FLOPs, but
                    # these layers cost the latency of their K loop, not arithmetic
                    cands["halo"] = lambda: nat.conv2d(x, conv.weight, conv.bias, stride=conv.stride[0], padding=conv.padding[0],
                                                       dilation=1, relu=relu, variant=7)
                if self._image2_ok(conv, x):
                    # round 6: conv6_2 (19 x 19 -> 10 x 10, stride 2) with the image resident in LDS, the strided taps as addresses, one
                    # (image, 64 channels) tile of 128 pixels per workgroup: 256 tiles at batch 32 instead of a split-K launch + its reduction
                    cands["image"] = lambda: nat.conv2d_image(x, conv.weight, conv.bias, stride=conv.stride[0], padding=conv.padding[0],
                                                              dilation=conv.dilation[0], relu=relu)
                    if (os.environ.get("SSDHIP_CONV", "auto") in ("auto", "auto_miopen") and not os.environ.get("SSDHIP_PREFER")
                            and x.shape[0] >= 16):
                        return cands["image"]()
            import os
            if ("splitk" in cands and self._splitk_measured_regime(x) and os.environ.get("SSDHIP_NO_SPLITK", "0") != "1"
                    and os.environ.get("SSDHIP_CONV", "auto") in ("auto", "auto_miopen") and not os.environ.get("SSDHIP_PREFER")):
                # The few-tile layers take the split-K form without a timing run: a back-to-back microbenchmark of these 5-30 us
                # kernels is host-bound and L2-warm and says nothing about them inside the step, where the form was measured
                # (r03p, graphed step, two A/B pairs: 2.435 -> 2.353 and 2.455 -> 2.378 ms; chain of extra layers 178 -> 138 us)
                return cands["splitk"]()
            name = (self._pick(("act", tuple(x.shape), conv.out_channels, k, conv.dilation[0], relu, conv.stride[0], conv.padding[0]), cands)
                    if len(cands) > 1 else "miopen")
            return cands[name]()
        if self._fused_train(x, conv):
            run, _name = self._train_thunk(conv, x, relu)
            if run is not None:
                wb, bb, wt = self._bf16_shadow(conv, with_transposed=True)
                return _ConvBiasActFn.apply(x, conv.weight, conv.bias, run, conv.stride, conv.padding, conv.dilation, relu, wb, bb, wt,
                                            link_in, link_out)
        y = conv(x)
        return F.relu(y) if relu else y

    # -- training step: the forward of every convolution libssdhip has a kernel for runs there (under autograd, see
    #    _ConvBiasActFn); pooling stays a separate PyTorch op because its backward needs the pre-pool activation ---------------
    def _fused_train(self, x, conv):
        return (self.fused_training and x.is_cuda and torch.is_grad_enabled() and conv.bias is not None and conv.groups == 1
                and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)))

    @staticmethod
    def _own_dgrad_ok(conv):
        """The data gradient of this layer runs on libssdhip's forward kernels with transposed / flipped filters (see
        _conv_input_weight_grads): stride 1, 'same', k in (1, 3), channel counts multiples of 64."""
        k = conv.kernel_size[0]
        if conv.groups != 1 or conv.kernel_size[1] != k or conv.in_channels % 64 or conv.out_channels % 64:
            return False
        if (k == 3 and conv.dilation == (1, 1) and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
                and conv.padding[0] in (0, 1) and isinstance(conv.padding, tuple)):
            return True                                  # (round 6) strided / 'valid' 3 x 3: embedding launch + the 'same' kernel
        return (conv.stride == (1, 1) and k in (1, 3) and conv.dilation[0] == conv.dilation[1]
                and conv.padding == (conv.dilation[0] * (k // 2),) * 2)

    def _shadow_build(self, device):
        """bf16 copies of every convolution's float32 master weights in the layouts the MFMA kernels read (csrc/ssdhip_optim.hip):
        channels_last filters, their transposed / tap-flipped twin where the layer's data gradient runs on our kernels, the bias; the
        conf and loc heads of a source map as ROWS OF ONE packed tensor ([conf | loc | zero rows up to a multiple of 128]: what
        _PackedHeadFn multiplies), so that nothing is concatenated, flipped or re-laid-out per step."""
        convs = [m for m in self.modules() if isinstance(m, nn.Conv2d) and m.bias is not None]
        index = {id(c): i for i, c in enumerate(convs)}
        cl, tr, bias, tr_arg = [None] * len(convs), [None] * len(convs), [None] * len(convs), [(0, 0)] * len(convs)
        packs = {}
        heads = list(zip(getattr(self, "conf_heads", []), getattr(self, "loc_heads", [])))
        with torch.no_grad():
            for l, (ch, lh) in enumerate(heads):
                same = lambda c: (isinstance(c, nn.Conv2d) and c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1)
                                  and c.dilation == (1, 1) and c.groups == 1 and c.bias is not None and id(c) in index)
                if not (same(ch) and same(lh) and ch.in_channels == lh.in_channels and ch.in_channels % 128 == 0):
                    continue
                nc, nl, cin = ch.out_channels, lh.out_channels, ch.in_channels
                cp = -(-(nc + nl) // 128) * 128
                pw = torch.zeros((cp, cin, 3, 3), dtype=torch.bfloat16, device=device).contiguous(memory_format=torch.channels_last)
                pwt = torch.zeros((cin, cp, 3, 3), dtype=torch.bfloat16, device=device).contiguous(memory_format=torch.channels_last)
                pb = torch.zeros((cp,), dtype=torch.bfloat16, device=device)
                packs[l] = (pw, pb, pwt, nc, nl)
                for conv, lo, n in ((ch, 0, nc), (lh, nc, nl)):
                    i = index[id(conv)]
                    cl[i], bias[i], tr[i], tr_arg[i] = pw[lo:lo + n], pb[lo:lo + n], pwt, (cp, lo)
            for i, c in enumerate(convs):
                if cl[i] is not None:
                    continue
                cl[i] = torch.empty(tuple(c.weight.shape), dtype=torch.bfloat16, device=device).contiguous(memory_format=torch.channels_last)
                if cl[i].dim() == 4 and not cl[i].permute(0, 2, 3, 1).is_contiguous():          # size-1 dims can leave odd strides behind
                    cl[i] = torch.empty(tuple(c.weight.permute(0, 2, 3, 1).shape), dtype=torch.bfloat16, device=device).permute(0, 3, 1, 2)
                bias[i] = torch.empty((c.out_channels,), dtype=torch.bfloat16, device=device)
                if self._own_dgrad_ok(c):
                    o, ci, kh, kw = c.weight.shape
                    tr[i] = torch.empty((ci, kh, kw, o), dtype=torch.bfloat16, device=device).permute(0, 3, 1, 2)   # (I, O, kh, kw) channels_last
                    tr_arg[i] = (o, 0)
        dests, seen = [], set()                                   # every tensor the refresh launch writes, once (views share a counter)
        for t in cl + tr + bias + [u for pk in packs.values() for u in pk[:3]]:
            if t is None:
                continue
            base = t._base if t._base is not None else t
            if id(base) not in seen:
                seen.add(id(base))
                dests.append(base)
        return {"device": device, "convs": convs, "index": index, "cl": cl, "tr": tr, "bias": bias, "tr_arg": tr_arg, "packs": packs,
                "key": None, "table": None, "table_key": None, "dests": tuple(dests)}

    def _shadow_state_fresh(self, conv):
        """The shadow state with every bf16 copy up to date (ONE launch over all parameters when any of them changed: the optimizer's
        in-place update bumps `_version`)."""
        st = self.__dict__.get("_shadow_state")
        if st is None or st["device"] != conv.weight.device:
            st = self._shadow_build(conv.weight.device)
            self.__dict__["_shadow_state"] = st
        # Inside raw_predictions the check runs once per forward pass (`_shadow_fresh`); a direct call of features() / conv_act()
        # checks every time.  The key holds the Parameter OBJECT, its storage and its version: an optimizer step bumps the version,
        # `param.data = t` changes the storage, `conv.weight = nn.Parameter(...)` the object.
        if not (self.__dict__.get("_in_forward", False) and self.__dict__.get("_shadow_fresh", False)):
            src = [c.weight for c in st["convs"]] + [c.bias for c in st["convs"]]
            key = tuple((id(t), t.data_ptr(), t._version) for t in src)
            if key != st["key"]:
                tkey = tuple((id(t), t.data_ptr(), tuple(t.stride())) for t in src)
                if st["table"] is None or st["table_key"] != tkey:             # the table holds raw pointers: rebuilt when a storage moved
                    w = [(c.weight.detach(), st["cl"][i], st["tr"][i], st["tr_arg"][i][0], st["tr_arg"][i][1]) for i, c in enumerate(st["convs"])]
                    v = [(c.bias.detach().contiguous(), st["bias"][i]) for i, c in enumerate(st["convs"])]
                    if any(b.data_ptr() != c.bias.data_ptr() for (b, _), c in zip(v, st["convs"])):
                        raise RuntimeError("a convolution bias is not a contiguous tensor")
                    st["table"] = nat.shadow_table(w, v, st["device"])
                    st["table_key"] = tkey
                nat.shadow_refresh(st["table"])
                # the kernel rewrote the shadows in place behind autograd's back: their version counters move as an in-place tensor
                # op's would, so a backward whose forward saved the OLD filters raises instead of multiplying the new ones
                _bump_versions(st["dests"])
                # Inside a stream capture the refresh is only RECORDED (it runs at every replay): the shadows are not fresh for the next
                # eager call, which must refresh them itself (tests/test_train_graph_gpu.py: the first eager step after a capture
                # multiplied the previous step's filters, 1.3e-3 off on the loss).
                if not torch.cuda.is_current_stream_capturing():
                    st["key"] = key
            self.__dict__["_shadow_fresh"] = True
        return st

    def _bf16_shadow(self, conv, with_transposed=False):
        """bf16 copies of a convolution's float32 master weight (channels_last) and bias for the libssdhip kernels [+ the transposed,
        tap-flipped filters of its data gradient, or None]; (None, None[, None]) for a convolution the shadow set does not hold."""
        if conv.weight.dtype == torch.bfloat16:
            return (conv.weight.detach(), conv.bias.detach(), None) if with_transposed else (conv.weight.detach(), conv.bias.detach())
        st = self._shadow_state_fresh(conv)
        i = st["index"].get(id(conv))
        if i is None:
            return (None, None, None) if with_transposed else (None, None)
        if with_transposed:
            packed = st["tr_arg"][i][0] != conv.out_channels               # a head's rows live in its source map's packed tensor
            return st["cl"][i], st["bias"][i], (None if packed else st["tr"][i])
        return st["cl"][i], st["bias"][i]

    def _packed_head_shadow(self, l):
        """(packed filters (Cp, Cin, 3, 3) bf16 channels_last, packed bias (Cp,), transposed / flipped (Cin, Cp, 3, 3), n_conf, n_loc) of
        predictor layer l, up to date; None when the heads of that source map are not packed."""
        ch = self.conf_heads[l]
        if ch.weight.dtype == torch.bfloat16:
            return None
        return self._shadow_state_fresh(ch)["packs"].get(l)

    def _train_thunk(self, conv, x, relu):
        """The libssdhip kernel for this layer as `(x_bf16, w_bf16, b_bf16) -> y`, or (None, None).  The variant is the one the
        per-shape autotune keeps (same keys as the inference path)."""
        k = conv.kernel_size[0]
        d = conv.dilation[0]
        if (conv.in_channels == 3 and conv.out_channels == 64 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
                and conv.padding == (1, 1) and conv.dilation == (1, 1)):
            return (lambda xb, wb, bb: nat.conv3x3_cin3(xb, wb, bb, relu=relu)), "cin3"
        if self._igemm_ok(conv, x):
            cands = {"igemm": lambda xb, wb, bb: nat.conv2d_same(xb, wb, bb, dilation=d, relu=relu),
                     "igemm6": lambda xb, wb, bb: nat.conv2d_same(xb, wb, bb, dilation=d, relu=relu, variant=6)}
            if conv.in_channels == 64 and k == 3 and d == 1:
                cands["c64"] = lambda xb, wb, bb: nat.conv3x3_c64(xb, wb, bb, relu=relu, pool=False)
            if self._halo_ok(conv, x):
                cands["halo"] = lambda xb, wb, bb: nat.conv2d_same(xb, wb, bb, dilation=1, relu=relu, variant=7)
            if self._image_ok(conv, x):
                cands["image"] = lambda xb, wb, bb: nat.conv3x3_image(xb, wb, bb, dilation=d, relu=relu)
            if k == 1 and self._image2_ok(conv, x):
                cands["image"] = lambda xb, wb, bb: nat.conv2d_image(xb, wb, bb, relu=relu)
        elif self._igemm_general_ok(conv, x):
            cands = {nm: (lambda xb, wb, bb, v=v: nat.conv2d(xb, wb, bb, stride=conv.stride[0], padding=conv.padding[0], dilation=d,
                                                            relu=relu, variant=v))
                     for nm, v in (("igemm", None), ("igemm5", 5), ("igemm6", 6))}
            if self._image2_ok(conv, x):
                cands["image"] = lambda xb, wb, bb: nat.conv2d_image(xb, wb, bb, stride=conv.stride[0], padding=conv.padding[0], dilation=d,
                                                                     relu=relu)
        else:
            return None, None
        import os
        key = ("act", tuple(x.shape), conv.out_channels, k, d, relu, conv.stride[0], conv.padding[0])
        forced = os.environ.get("SSDHIP_CONV", "auto")
        if len(cands) == 1:
            name = "igemm"
        elif forced in cands:
            name = forced
        elif SSDModel._conv_choice.get(key) in cands:
            name = SSDModel._conv_choice[key]                   # settled earlier: no bf16 copies of x / weight / bias just to look it up
        else:
            with torch.no_grad():
                xb = x.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                wb = conv.weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                bb = conv.bias.detach().to(torch.bfloat16)
                name = self._pick(key, {n: (lambda fn=fn: fn(xb, wb, bb)) for n, fn in cands.items()})
            if name not in cands:
                name = "igemm"
        return cands[name], name

    @staticmethod
    def relu_link():
        """A link between two consecutive ReLU convolutions of the training step (the lower one's output feeds ONLY the upper one):
        pass it as `link_out` of the lower layer's conv_act and `link_in` of the upper layer's conv_act / conv_act_pool.  None outside
        autograd -- the inference paths take no links."""
        return _ReluLink() if torch.is_grad_enabled() else None

    def conv_act_pool(self, conv, x, kernel, stride, pad=0, ceil_mode=False, link_in=None):
        if self._fused(x, conv):
            cands = {"miopen": lambda: nat.bias_act_maxpool(self._conv_nobias(conv, x), conv.bias, kernel, stride, pad, ceil_mode,
                                                            relu=True)}
            if self._igemm_ok(conv, x):
                cands["igemm"] = lambda: nat.bias_act_maxpool(
                    nat.conv2d_same(x, conv.weight, conv.bias, dilation=conv.dilation[0], relu=True), None, kernel, stride, pad,
                    ceil_mode, relu=False)
                cands["igemm6"] = lambda: nat.bias_act_maxpool(
                    nat.conv2d_same(x, conv.weight, conv.bias, dilation=conv.dilation[0], relu=True, variant=6), None, kernel, stride,
                    pad, ceil_mode, relu=False)
                if self._halo_ok(conv, x):               # the slab kernel + a separate pooling pass can beat the fused epilogue
                    cands["halo"] = lambda: nat.bias_act_maxpool(
                        nat.conv2d_same(x, conv.weight, conv.bias, dilation=1, relu=True, variant=7), None, kernel, stride, pad,
                        ceil_mode, relu=False)
                if self._image_ok(conv, x):              # conv5_3 -> pool5: the image-resident kernel + the pooling pass
                    cands["image"] = lambda: nat.bias_act_maxpool(
                        nat.conv3x3_image(x, conv.weight, conv.bias, dilation=conv.dilation[0], relu=True), None, kernel, stride, pad,
                        ceil_mode, relu=False)
                if (kernel == 2 and stride == 2 and pad == 0 and (ceil_mode or x.shape[2] % 2 == 0) and (ceil_mode or x.shape[3] % 2 == 0)):
                    # pooling fused into the convolution's epilogue: the full-resolution activation is never written
                    cands["igemm_pool"] = lambda: nat.conv2d_same_pool2(x, conv.weight, conv.bias, dilation=conv.dilation[0], relu=True)
                    if self._halo_ok(conv, x):
                        cands["halo_pool"] = lambda: nat.conv3x3_halo(x, conv.weight, conv.bias, relu=True, pool=True)
                    if conv.in_channels == 64 and conv.kernel_size == (3, 3) and conv.dilation[0] == 1:
                        cands["c64_pool"] = lambda: nat.conv3x3_c64(x, conv.weight, conv.bias, relu=True, pool=True)
            name = (self._pick(("pool", tuple(x.shape), conv.out_channels, conv.kernel_size[0], conv.dilation[0], kernel, stride, pad),
                               cands) if len(cands) > 1 else "miopen")
            return cands[name]()
        import os
        if (kernel == 2 and stride == 2 and pad == 0 and ceil_mode and self._fused_train(x, conv) and conv.out_channels % 8 == 0
                and 256 % (conv.out_channels // 8) == 0 and os.environ.get("SSDHIP_NO_FUSED_POOL_BWD", "0") != "1"):
            run, _name = self._train_thunk(conv, x, True)
            if run is not None:
                wb, bb, wt = self._bf16_shadow(conv, with_transposed=True)
                return _ConvBiasActPoolFn.apply(x, conv.weight, conv.bias, run, conv.stride, conv.padding, conv.dilation, wb, bb, wt, link_in)
        return self.max_pool(self.conv_act(conv, x, relu=True, link_in=link_in), kernel, stride, pad, ceil_mode=ceil_mode)

    def conv1_block_pool(self, c1, c2, x):
        """conv1_1 -> conv1_2 -> MaxPooling2D(2, 2, 'same') (models/keras_ssd300.py:274-276).  On the fused bf16 inference path the
        three can run as ONE kernel that never writes the 64-channel full-resolution map (csrc/ssdhip_conv64.hip, FRONT); timed once
        per shape against the two-kernel form."""
        same3 = lambda c: (c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1)
                           and c.groups == 1 and c.bias is not None)
        import os
        if (os.environ.get("SSDHIP_NO_CONV1_BLOCK", "0") != "1" and os.environ.get("SSDHIP_CONV", "auto") in ("auto", "auto_miopen")
                and self._fused(x, c1) and self._fused(x, c2) and same3(c1) and same3(c2)
                and c1.in_channels == 3 and c1.out_channels == 64
                and c2.in_channels == 64 and c2.out_channels % 64 == 0):
            cands = {"separate": lambda: self.conv_act_pool(c2, self.conv_act(c1, x), 2, 2, ceil_mode=True),
                     "conv1_block": lambda: nat.conv1_block(x, c1.weight, c1.bias, c2.weight, c2.bias, relu=True, pool=True)}
            key = ("conv1_block", tuple(x.shape), c2.out_channels)
            if SSDModel._conv_choice.get(key) is None:
                cands["separate"]()                          # settles the inner per-layer choices before the two forms are compared
            name = self._pick(key, cands)
            return cands[name if name in cands else "separate"]()       # a forced SSDHIP_CONV mode names per-layer kernels: two-kernel form
        return self.conv_act_pool(c2, self.conv_act(c1, x), 2, 2, ceil_mode=True)

    def max_pool(self, x, kernel, stride, pad=0, ceil_mode=False):
        if self._fused(x) and x.shape[1] % 8 == 0:
            return nat.bias_act_maxpool(x, None, kernel, stride, pad, ceil_mode, relu=False)
        import os
        if (self.fused_training and x.is_cuda and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and x.shape[1] % 8 == 0
                and x.requires_grad and os.environ.get("SSDHIP_NO_OWN_POOL", "0") != "1"):
            return _MaxPoolFn.apply(x, kernel, stride, pad, ceil_mode)
        return F.max_pool2d(x, kernel, stride, pad, ceil_mode=ceil_mode)

    # -- in-graph input pipeline (keras_ssd300.py:247-272): NHWC 0..255 -> normalised NCHW (channels_last memory) --
    def preprocess(self, images):
        x = images
        if x.dim() != 4:
            raise ValueError("expected images of shape (batch, height, width, channels)")
        nhwc = x.shape[-1] == self.img_channels and x.shape[1] != self.img_channels
        if (nhwc and self.fused_inference and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
                and not torch.is_grad_enabled() and self.img_channels <= 4
                and next(self.parameters()).dtype == torch.bfloat16):
            return nat.preprocess(x, self.subtract_mean, self.divide_by_stddev,
                                  list(self.swap_channels) if self.swap_channels else None)
        if (nhwc and self.fused_training and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and torch.is_grad_enabled()
                and not x.requires_grad and self.img_channels <= 4 and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16 and os.environ.get("SSDHIP_NO_TRAIN_PREPROCESS", "0") != "1"):
            # the training step under bf16 autocast: the first convolution casts its input to bf16 anyway, and the images need no
            # gradient -- the same one-launch kernel instead of a subtraction, an index_select and two layout / dtype copies
            return nat.preprocess(x, self.subtract_mean, self.divide_by_stddev,
                                  list(self.swap_channels) if self.swap_channels else None)
        if nhwc:
            x = x.permute(0, 3, 1, 2)                # NHWC storage == channels_last NCHW: no copy
        x = x.float()
        # the per-channel constants live on the device (built once): nothing crosses PCIe per call, and the step can be captured
        # into a HIP graph (a host -> device copy is not allowed while a stream is capturing)
        if self.subtract_mean is not None:
            x = x - self._device_const("mean", x.device, lambda: torch.as_tensor(self.subtract_mean, dtype=torch.float32).view(1, -1, 1, 1))
        if self.divide_by_stddev is not None:
            x = x / self._device_const("std", x.device, lambda: torch.as_tensor(self.divide_by_stddev, dtype=torch.float32).view(1, -1, 1, 1))
        if self.swap_channels:
            x = x.index_select(1, self._device_const("swap", x.device, lambda: torch.as_tensor(list(self.swap_channels), dtype=torch.long)))
        return x.contiguous(memory_format=torch.channels_last)

    def _device_const(self, name, device, build):
        key = ("const", name, str(device))
        t = self._anchor_cache.get(key)
        if t is None:
            t = build().to(device)
            self._anchor_cache[key] = t
        return t

    def anchors_and_variances(self, feature_sizes, device):
        """(N, 8) float32 constant, resident per device."""
        key = (tuple(feature_sizes), str(device))
        t = self._anchor_cache.get(key)
        if t is None:
            t = torch.cat([pb.constant(h, w, device).reshape(-1, 8) for pb, (h, w) in zip(self.priorboxes, feature_sizes)], dim=0)
            self._anchor_cache[key] = t
        return t

    def predictor_sizes(self):
        raise NotImplementedError

    def precise(self, enable=True, check_finite=True):
        """Make the REFERENCE's precision on the MFMA path this model's forward path (models/precise.py: float32 weights, every
        activation and filter a float16 (hi, lo) pair, three float16 MFMA products per multiplication, float32 accumulation; per-layer
        power-of-two range scales with an overflow guard): `model(images)` / `model.raw_predictions(images)` of a float32 CUDA model
        under no_grad then run there -- 2.6 x the framework's float32 convolutions at batch 32, predictions within ~1e-6 of them.
        `model.precise(False)` switches back.  Returns the model.  Reference: models/keras_ssd300.py:274-419 (a float32 graph)."""
        if not enable:
            self.__dict__["_precise"] = None
            return self
        from .precise import PreciseForward
        self.__dict__["_precise"] = PreciseForward(self, check_finite=check_finite)
        return self

    def raw_predictions(self, images, decode=False):
        """The `(batch, #boxes, #classes + 12)` prediction tensor; with `decode=True` (used by `forward` in the inference modes) the
        decoded detections, which on the fused bf16 path come straight from the head outputs."""
        pf = self.__dict__.get("_precise")
        if (pf is not None and images.is_cuda and not torch.is_grad_enabled() and next(self.parameters()).dtype == torch.float32):
            return pf(images, decode=bool(decode and self.decoder is not None))
        self.__dict__["_shadow_fresh"] = False
        self.__dict__["_in_forward"] = True
        try:
            return self._raw_predictions(images, decode)
        finally:
            self.__dict__["_in_forward"] = False

    def _raw_predictions(self, images, decode):
        x = self.preprocess(images)
        dtype = next(self.parameters()).dtype
        xin = x.to(dtype) if not torch.is_autocast_enabled() else x
        split = self._split_heads(xin)
        if split is not None:
            feats, confs = split
            sizes = [(f.shape[2], f.shape[3]) for f in feats]
            anchors = self.anchors_and_variances(sizes, x.device)
            head_args = (confs, [None] * len(confs), [ch.bias for ch in self.conf_heads], [lh.bias for lh in self.loc_heads],
                         [pb.n_boxes for pb in self.priorboxes], anchors, self.n_classes)
            if decode and self.decoder is not None and self.n_classes <= 81:
                return self.decoder.forward_from_heads(*head_args)
            pred = nat.assemble_predictions(*head_args)
            return self.decoder(pred) if (decode and self.decoder is not None) else pred
        feats = self.features(xin)
        b = x.shape[0]
        sizes = [(f.shape[2], f.shape[3]) for f in feats]
        if all(self._fused_head_ok(f, ch) for f, ch in zip(feats, self.conf_heads)):
            # heads without bias; bias, Reshape, softmax, anchors and the concatenations are one libssdhip pass (:363-419)
            packable = all(self._packed_head_ok(ch, lh, f) for f, ch, lh in zip(feats, self.conf_heads, self.loc_heads))
            how = "per_layer"
            if packable and len(feats) <= 8:
                # all packed heads as ONE grouped launch vs one launch per layer: the 1x1..5x5 heads are single workgroups
                # walking long K loops (pure latency); side by side they hide behind the 38x38 / 19x19 heads
                hc = {"per_layer": lambda: self._heads_per_layer(feats), "grouped": lambda: self._heads_grouped(feats)}
                if self._halo_heads_ok(feats):
                    hc["halo_grouped"] = lambda: self._heads_halo_grouped(feats)
                elif self._halo_heads_mixed_ok(feats):
                    hc["halo_mixed"] = lambda: self._heads_halo_mixed(feats)
                how = self._pick(("heads", tuple(tuple(f.shape) for f in feats), self.n_classes), hc)
            confs, locs = (self._heads_halo_grouped(feats) if how == "halo_grouped" else
                           self._heads_halo_mixed(feats) if how == "halo_mixed" else
                           self._heads_grouped(feats) if how == "grouped" else self._heads_per_layer(feats))
            anchors = self.anchors_and_variances(sizes, x.device)
            head_args = (confs, locs, [ch.bias for ch in self.conf_heads], [lh.bias for lh in self.loc_heads],
                         [pb.n_boxes for pb in self.priorboxes], anchors, self.n_classes)
            if decode and self.decoder is not None and self.n_classes <= 81:
                return self.decoder.forward_from_heads(*head_args)          # y_pred is never materialised
            pred = nat.assemble_predictions(*head_args)
            return self.decoder(pred) if (decode and self.decoder is not None) else pred
        import os
        if (torch.is_grad_enabled() and not decode and os.environ.get("SSDHIP_NO_TRAIN_ASSEMBLY", "0") != "1" and len(feats) <= 8
                and all(self._packed_train_head_ok(f, ch, lh) for f, ch, lh in zip(feats, self.conf_heads, self.loc_heads))
                and self._train_assembly_fits(len(feats))):
            # every source map's heads are one packed libssdhip node: the assembly and its backward are one launch each (the backward
            # kernel stages two 128-anchor row tiles + a packed stage in LDS; whether that fits is the kernel's own formula,
            # nat.assemble_backward_supported -- 21 classes need 36 KB, COCO's 81 classes 117 KB of the CU's 160)
            ys = []
            for l, (f, ch, lh) in enumerate(zip(feats, self.conf_heads, self.loc_heads)):
                pw, pb, pwt, _nc, _nl = self._packed_head_shadow(l)
                ys.append(_PackedHeadFn.apply(f, ch.weight, ch.bias, lh.weight, lh.bias, pw, pb, pwt))
            anchors = self.anchors_and_variances(sizes, x.device)
            return _AssembleTrainFn.apply(anchors, self.n_classes, [pb_.n_boxes for pb_ in self.priorboxes], *ys)
        confs, locs = [], []
        for f, ch, lh in zip(feats, self.conf_heads, self.loc_heads):
            # NCHW -> NHWC before the reshape so the channel axis splits as (box, class) like Keras (:363-383)
            if self._packed_train_head_ok(f, ch, lh):
                pw, pb, pwt, _nc, _nl = self._packed_head_shadow(len(confs))
                y = _PackedHeadFn.apply(f, ch.weight, ch.bias, lh.weight, lh.bias, pw, pb, pwt).permute(0, 2, 3, 1)
                confs.append(y[..., :ch.out_channels].reshape(b, -1, self.n_classes))
                locs.append(y[..., ch.out_channels:ch.out_channels + lh.out_channels].reshape(b, -1, 4))
                continue
            confs.append(ch(f).permute(0, 2, 3, 1).reshape(b, -1, self.n_classes))
            locs.append(lh(f).permute(0, 2, 3, 1).reshape(b, -1, 4))
        conf = torch.softmax(torch.cat(confs, dim=1).float(), dim=-1)          # 'mbox_conf_softmax' (:415)
        loc = torch.cat(locs, dim=1).float()
        anchors = self.anchors_and_variances(sizes, conf.device)
        pred = torch.cat([conf, loc, anchors.unsqueeze(0).expand(b, -1, -1)], dim=2)   # 'predictions' (:419)
        return self.decoder(pred) if (decode and self.decoder is not None) else pred

    def _split_heads(self, x):
        """Fused bf16 inference only: the predictor heads of the trunk's two source maps (conv4_3, fc7: ~85 % of the head FLOPs) do not
        depend on the extra layers -- a chain of eight small convolutions that leaves most CUs idle -- so the two can share the chip
        on two HIP streams.  Mode 3 is what the HIP-graph step uses (GraphedInference; the eager path stays on one stream unless
        SSDHIP_HEAD_OVERLAP says otherwise): the two trunk heads as a grouped slab launch capped at 160 of the 256 CUs on the second
        stream beside the chain, then the four small heads (1.4 % / 0.7 % of a step in round 2; no measurable difference since the
        extra layers are split-K launches: profiles/r03zd_two_stream_heads_and_producer_priority_remeasured.txt).  Modes 1 | 2 are the older forms with the implicit-GEMM heads (1: the heads on the second stream; 2: the chain on
        a high-priority second stream): ~190 us of kernels run side by side but slow each other down by as much -- those heads fill
        every CU (r02p: 2.831 off / 2.832 / 2.815 ms).  Returns (feature maps, packed head outputs), or None for the one-stream path."""
        import os
        mode = os.environ.get("SSDHIP_HEAD_OVERLAP") or self.__dict__.get("_head_overlap") or "0"
        if (mode == "0" or not hasattr(self, "trunk_features") or not x.is_cuda
                or torch.is_grad_enabled() or not self.fused_inference or x.dtype != torch.bfloat16
                or len(self.conf_heads) > 8 + 2):
            return None
        if not all(conv.weight.dtype == torch.bfloat16 for conv in self.conf_heads):
            return None
        early = self.trunk_features(x)
        n_early = len(early)
        if not all(self._fused_head_ok(f, ch) and self._packed_head_ok(ch, lh, f)
                   for f, ch, lh in zip(early, self.conf_heads, self.loc_heads)):
            raise RuntimeError("predictor heads of the trunk do not qualify for the packed kernel")
        if mode in ("3", "4") and not self._halo_heads_ok(early):
            mode = "1"
        main = torch.cuda.current_stream(x.device)
        side = self.__dict__.get("_side_stream")
        if side is None or side.device != x.device:
            # An ORDINARY stream since round 6 (rounds 2-5: priority -1, "served first when both streams have workgroups pending" --
            # schedule 4 has no such moment: the capped launch leaves the chain its CUs).  Same box, 6 x 60 steps alternating: 2.0068 ms
            # (ordinary) vs 2.0072 ms (high priority); but once a process has USED a high-priority stream, every later HIP graph with
            # parallel branches replays slower on this runtime -- the reference-precision step 6.5 -> 7.6-7.9 ms, slower than its eager
            # form, which is what bench.py's second graph measured in rounds 5-6 (profiles/r06z_graphs_after_a_high_priority_stream.txt).
            side = torch.cuda.Stream(device=x.device, priority=int(os.environ.get("SSDHIP_SIDE_PRIORITY", "0")))
            self.__dict__["_side_stream"] = side

        def check_rest(rest):
            if not all(self._fused_head_ok(f, ch) and self._packed_head_ok(ch, lh, f)
                       for f, ch, lh in zip(rest, self.conf_heads[n_early:], self.loc_heads[n_early:])):
                raise RuntimeError("predictor heads of the extra layers do not qualify for the packed kernel")

        # No record_stream anywhere: every tensor the other stream touches outlives the join in program order, and a block of the
        # side stream's pool is only reused after that stream has waited for the current one again.
        if mode == "4" and hasattr(self, "extra_features_front") and hasattr(self, "extra_features_tail"):
            # Round 4: the tail of the extra layers is ONE launch on one CU per image (csrc/ssdhip_chain.hip: 32 CUs, ~55 us), so the
            # balance moved: first the front of the extra layers (conv6_1, conv6_2: split-K launches that want the whole chip), THEN
            # the two trunk heads on the second stream capped so that one CU per image stays free, beside the tail and the small heads
            front = self.extra_features_front(early[1])
            # Round 6: conv6_2 exists BEFORE the fork, so its head rides in the capped launch with the two trunk heads instead of leading
            # the small launch behind the chain.  In units of 72 K-steps the capped launch is then 112 fc7 items x 2 + 192 conv4_3 items
            # + 32 conv6_2 items = 448 = exactly two per workgroup at 224 workgroups in the kernel's snake order (at 216 sixteen
            # workgroups draw an fc7 item AND a conv6_2 item: 132 us instead of 102), and 224 + the chain's 32 = the chip's 256 CUs.
            # Same box, alternating, 3 x 30 steps: 1.935 -> 1.926 ms per step (profiles/r06y_ab_head_split.txt).
            n_big = n_early + 1 if (os.environ.get("SSDHIP_HEAD_SPLIT", "3") == "3" and self._halo_heads_ok([front])
                                    and self._fused_head_ok(front, self.conf_heads[n_early])
                                    and self._packed_head_ok(self.conf_heads[n_early], self.loc_heads[n_early], front)) else n_early
            big_maps = list(early) + ([front] if n_big > n_early else [])
            side.wait_stream(main)
            with torch.cuda.stream(side):
                big = nat.conv3x3_halo_group(big_maps, [self._packed_head_weight(l, 128) for l in range(n_big)], None, relu=False,
                                             max_workgroups=int(os.environ.get("SSDHIP_HEAD_WGS", "224" if n_big > n_early else "216")))
            rest = self.extra_features_tail(front)
            check_rest(rest)
            later = rest[n_big - n_early:]                    # the maps whose heads are still to come
            if not later:
                small = []
            elif self._halo_heads_ok(later):
                small = nat.conv3x3_halo_group(list(later), [self._packed_head_weight(n_big + l, 128) for l in range(len(later))], None,
                                               relu=False)
            else:
                small = nat.conv2d_same_group(list(later), [self._packed_head_weight(n_big + l) for l in range(len(later))], None,
                                              relu=False)
            main.wait_stream(side)
            return early + rest, big + small
        if mode in ("3", "4"):
            # the two trunk heads as a grouped slab launch capped at HALF the CUs (persistent workgroups, one per CU) on the second
            # stream, the latency-bound chain of extra layers on the current one in the other half, then the four small heads
            side.wait_stream(main)
            with torch.cuda.stream(side):
                big = nat.conv3x3_halo_group(list(early), [self._packed_head_weight(l, 128) for l in range(n_early)], None, relu=False,
                                             max_workgroups=int(os.environ.get("SSDHIP_HEAD_WGS", "128")))
            rest = self.extra_features(early[1])
            check_rest(rest)
            if self._halo_heads_ok(rest):
                small = nat.conv3x3_halo_group(list(rest), [self._packed_head_weight(n_early + l, 128) for l in range(len(rest))], None,
                                               relu=False)
            else:                                            # extra maps the slab kernel does not cover: the implicit-GEMM group
                small = nat.conv2d_same_group(list(rest), [self._packed_head_weight(n_early + l) for l in range(len(rest))], None,
                                              relu=False)
            main.wait_stream(side)
            return early + rest, big + small
        if mode == "2":
            # the latency-bound chain (extra layers + their small heads) on the high-priority stream, the two big heads on the current
            # one: the chain's few workgroups no longer queue behind ~600 head workgroups at every one of its eight launches
            side.wait_stream(main)
            with torch.cuda.stream(side):
                rest = self.extra_features(early[1])
                check_rest(rest)
                small = nat.conv2d_same_group(list(rest), [self._packed_head_weight(n_early + l) for l in range(len(rest))], None,
                                              relu=False)
            big = nat.conv2d_same_group(list(early), [self._packed_head_weight(l) for l in range(n_early)], None, relu=False)
            main.wait_stream(side)
            return early + rest, big + small
        side.wait_stream(main)
        with torch.cuda.stream(side):
            big = nat.conv2d_same_group(list(early), [self._packed_head_weight(l) for l in range(n_early)], None, relu=False)
        rest = self.extra_features(early[1])
        check_rest(rest)
        small = nat.conv2d_same_group(list(rest), [self._packed_head_weight(n_early + l) for l in range(len(rest))], None, relu=False)
        main.wait_stream(side)
        return early + rest, big + small

    def _train_assembly_fits(self, n_maps):
        """The one-launch assembly backward runs for this model's packed heads (LDS need from libssdhip's own formula, cached)."""
        packs = [self._packed_head_shadow(l) for l in range(n_maps)]
        if any(pk is None for pk in packs):
            return False
        key = (self.n_classes, tuple(int(pb.n_boxes) for pb in self.priorboxes[:n_maps]), tuple(int(pk[0].shape[0]) for pk in packs))
        memo = self.__dict__.setdefault("_assembly_fits", {})
        if key not in memo:
            memo[key] = nat.assemble_backward_supported(key[0], key[1], key[2])
        return memo[key]

    def _packed_train_head_ok(self, f, ch, lh):
        """Training step, bf16 autocast on the GPU, 3x3 'same' heads on a map the slab / weight-gradient kernels cover."""
        import os
        return (self._fused_train(f, ch) and lh.bias is not None and self._packed_head_ok(ch, lh, f) and f.shape[1] % 128 == 0
                and f.shape[3] <= 190 and os.environ.get("SSDHIP_NO_OWN_HEADS", "0") != "1"
                and ch in self.conf_heads and self._packed_head_shadow(list(self.conf_heads).index(ch)) is not None)

    def _heads_grouped(self, feats):
        outs = nat.conv2d_same_group(list(feats), [self._packed_head_weight(l) for l in range(len(feats))], None, relu=False)
        return outs, [None] * len(outs)

    def _heads_halo_grouped(self, feats):
        """All packed heads through the slab kernel in one launch of persistent workgroups (csrc/ssdhip_convh.hip): filters padded to a
        multiple of 128 output channels; the deepest head (fc7's: 144 K-steps) is dispatched first."""
        outs = nat.conv3x3_halo_group(list(feats), [self._packed_head_weight(l, 128) for l in range(len(feats))], None, relu=False)
        return outs, [None] * len(outs)

    def _heads_halo_mixed(self, feats):
        """Round 6, fourth session (SSD512: its conv4_3 map is 64 wide, two columns more than the grouped slab launch's LDS layout takes,
        and ALL seven heads fell back to the implicit-GEMM group -- 223 us of a 3.0 ms step): the maps wider than 62 each through the
        single-problem slab entry (which tiles them as it sees fit: 16 x 16-pixel tiles, one round of 256 workgroups at batch 16), the
        others as the grouped slab launch."""
        wide = [l for l, f in enumerate(feats) if f.shape[3] > 62]
        rest = [l for l in range(len(feats)) if l not in wide]
        outs = [None] * len(feats)
        for l in wide:
            outs[l] = nat.conv3x3_halo(feats[l], self._packed_head_weight(l, 128), None, relu=False, pool=False)
        if rest:
            got = nat.conv3x3_halo_group([feats[l] for l in rest], [self._packed_head_weight(l, 128) for l in rest], None, relu=False)
            for l, y in zip(rest, got):
                outs[l] = y
        return outs, [None] * len(outs)

    def _halo_heads_mixed_ok(self, feats):
        import os
        rest = [f for f in feats if f.shape[3] <= 62]
        return (os.environ.get("SSDHIP_NO_HALO", "0") != "1" and os.environ.get("SSDHIP_NO_HALO_MIXED", "0") != "1"
                and len(rest) <= 8 and len(rest) < len(feats)
                and all(f.shape[1] % 128 == 0 for f in feats))

    def _halo_heads_ok(self, feats):
        import os
        return (os.environ.get("SSDHIP_NO_HALO", "0") != "1" and len(feats) <= 8
                and all(f.shape[1] % 128 == 0 and f.shape[3] <= 62 for f in feats))

    def _heads_per_layer(self, feats):
        """Per layer the two heads run either as two MIOpen convolutions or PACKED into one libssdhip implicit-GEMM launch
        (conf and loc filters concatenated along Cout, zero-padded to 64 channels): timed once per shape, faster kept."""
        confs, locs = [], []
        for l, (f, ch, lh) in enumerate(zip(feats, self.conf_heads, self.loc_heads)):
            cands = {"miopen": lambda f=f, ch=ch, lh=lh: (self._conv_nobias(ch, f), self._conv_nobias(lh, f))}
            if self._packed_head_ok(ch, lh, f):
                cands["igemm"] = lambda f=f, l=l: (nat.conv2d_same(f, self._packed_head_weight(l), None, dilation=1, relu=False), None)
            name = self._pick(("head", l, tuple(f.shape), ch.out_channels, lh.out_channels), cands) if len(cands) > 1 else "miopen"
            c, lo = cands[name]()
            confs.append(c)
            locs.append(lo)
        return confs, locs

    @staticmethod
    def _packed_head_ok(ch, lh, f):
        same = lambda c: (c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1) and c.groups == 1)
        return same(ch) and same(lh) and ch.in_channels % 64 == 0 and ch.in_channels == lh.in_channels

    def _packed_head_weight(self, l, multiple=64):
        """[conf filters | loc filters | zero rows up to a multiple of `multiple`] of predictor layer l as one (Cout, Cin, 3, 3) bf16
        weight in channels_last memory; rebuilt when either head's weight tensor changes (in-place updates bump `_version`)."""
        ch, lh = self.conf_heads[l], self.loc_heads[l]
        key = (ch.weight._version, lh.weight._version, ch.weight.data_ptr(), lh.weight.data_ptr())
        hit = self._packed_heads.get((l, multiple))
        n = ch.out_channels + lh.out_channels
        if hit is not None and hit[0] != key and hit[1].device == ch.weight.device and hit[1].dtype == ch.weight.dtype:
            # refreshed IN PLACE: a captured HIP graph (GraphedInference) keeps reading this storage
            with torch.no_grad():
                hit[1][:ch.out_channels].copy_(ch.weight)
                hit[1][ch.out_channels:n].copy_(lh.weight)
            hit = (key, hit[1])
            self._packed_heads[(l, multiple)] = hit
        elif hit is None or hit[0] != key:
            pad = (-n) % multiple
            with torch.no_grad():
                w = torch.cat([ch.weight, lh.weight] + ([ch.weight.new_zeros((pad,) + tuple(ch.weight.shape[1:]))] if pad else []), dim=0)
                w = w.contiguous(memory_format=torch.channels_last)
            hit = (key, w)
            self._packed_heads[(l, multiple)] = hit
        return hit[1]

    def _head_weights_key(self):
        return tuple((c.weight._version, c.weight.data_ptr()) for heads in (self.conf_heads, self.loc_heads) for c in heads)

    def _refresh_packed_heads(self):
        """Rebuild every cached packed head filter IN ITS OWN STORAGE (a captured HIP graph keeps reading that storage)."""
        for (l, multiple) in list(self._packed_heads):
            self._packed_head_weight(l, multiple)

    def _derived_weights_key(self):
        """Versions of every parameter some cached, derived tensor was built from (what a captured graph cannot follow by itself)."""
        gammas = tuple((m.gamma.data_ptr(), m.gamma._version) for m in self.modules()
                       if isinstance(m, L2Normalization) and m.gamma is not None)
        return self._head_weights_key() + gammas

    def _refresh_derived_weights(self):
        self._refresh_packed_heads()
        for m in self.modules():
            if isinstance(m, L2Normalization):
                m.refresh_cached_gamma()

    def _fused_head_ok(self, f, conv):
        return (self.fused_inference and f.is_cuda and f.dtype == torch.bfloat16 and not torch.is_grad_enabled()
                and conv.weight.dtype == torch.bfloat16)

    def forward(self, images):
        return self.raw_predictions(images, decode=self.decoder is not None)

    def head_outputs(self, images):
        """The convolution stack alone: the in-graph input pipeline, every trunk / extra layer and the packed predictor heads -- what
        `forward` runs in front of DecodeDetections (or of the prediction assembly), nothing behind it.  Returns the list of packed
        head maps (one per source map, conf | loc channels).  bench.py times a HIP graph of it for `conv_roofline.forward_ms`."""
        if torch.is_grad_enabled():
            raise RuntimeError("head_outputs is an inference-path probe: call it under torch.no_grad()")
        self.__dict__["_shadow_fresh"] = False
        self.__dict__["_in_forward"] = True
        had = self.__dict__.get("_head_overlap")
        if not had:
            self.__dict__["_head_overlap"] = "4"           # outside a graph capture too: the graph step's two-stream schedule
        try:
            x = self.preprocess(images)
            split = self._split_heads(x.to(next(self.parameters()).dtype))
            if split is None:
                raise RuntimeError("head_outputs: this model / input does not take the fused inference path")
            return list(split[1])
        finally:
            self.__dict__["_head_overlap"] = had
            self.__dict__["_in_forward"] = False

    def graphed(self, images, warmup=3, heads_only=False):
        """`forward` for inputs of this shape captured ONCE into a HIP graph: a step is then a single graph launch instead of ~45
        kernel launches issued from Python (the step is 3 ms of GPU work; on a slow or busy host the eager launches alone can take
        longer).  Returns a callable; see GraphedInference.  `heads_only`: the graph of `head_outputs` (the convolution stack with the
        same two-stream schedule, no decode)."""
        return GraphedInference(self, images, warmup, fn=self.head_outputs if heads_only else None)

    predict = forward

    def l2_regularization_loss(self):
        """Keras adds l2(l2_reg) * sum(W^2) over every conv kernel (not biases) to the loss (:274 kernel_regularizer)."""
        if not self.l2_regularization:
            return torch.zeros((), device=next(self.parameters()).device)
        return self.l2_regularization * sum((m.weight.float() ** 2).sum() for m in self.modules() if isinstance(m, nn.Conv2d))


def make_priorboxes(img_height, img_width, scales, aspect_ratios, two_boxes_for_ar1, steps, offsets, clip_boxes,
                    variances, coords, normalize_coords, names):
    return nn.ModuleList([
        AnchorBoxes(img_height, img_width, this_scale=scales[i], next_scale=scales[i + 1], aspect_ratios=aspect_ratios[i],
                    two_boxes_for_ar1=two_boxes_for_ar1, this_steps=steps[i], this_offsets=offsets[i],
                    clip_boxes=clip_boxes, variances=variances, coords=coords, normalize_coords=normalize_coords,
                    name=names[i])
        for i in range(len(names))])
