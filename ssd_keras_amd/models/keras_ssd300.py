"""`ssd_300` -- drop-in for the reference builder models/keras_ssd300.py:31-457, as a torch module.

VGG-16 (atrous fc6/fc7) + extra feature layers + 6 pairs of 3x3 predictor heads; 8732 anchors at
300x300.  The modules are plain `nn.Conv2d` containers of the parameters; what multiplies depends on the
dtype: a bf16 channels_last model on a GPU runs every convolution on libssdhip's hand-written MFMA kernels
(models/_common.py picks per layer shape: the slab kernel, the fused conv1 block, the image-resident and
implicit-GEMM kernels, the extra-layer chain), the training step the same kernels and their gradients under
autograd, `model.precise()` their float16 x 3 forms at float32 precision; a float32 model is the framework's own
convolution (MIOpen), kept as the numerics reference.  Layer names follow the reference so ported weights can
be loaded by name.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .. import _native as nat

from ..keras_layers.keras_layer_L2Normalization import L2Normalization
from ._common import SSDModel, conv_out, he_normal_, make_priorboxes, pool_out, resolve_anchor_config


class _VGGBase(SSDModel):
    """conv1_1 .. fc7 shared by SSD300 and SSD512 (keras_ssd300.py:274-300)."""

    def _build_vgg(self, in_ch):
        def c(i, o, k=3, **kw):
            return nn.Conv2d(i, o, k, padding=kw.pop('padding', k // 2), **kw)
        self.conv1_1, self.conv1_2 = c(in_ch, 64), c(64, 64)
        self.conv2_1, self.conv2_2 = c(64, 128), c(128, 128)
        self.conv3_1, self.conv3_2, self.conv3_3 = c(128, 256), c(256, 256), c(256, 256)
        self.conv4_1, self.conv4_2, self.conv4_3 = c(256, 512), c(512, 512), c(512, 512)
        self.conv5_1, self.conv5_2, self.conv5_3 = c(512, 512), c(512, 512), c(512, 512)
        self.fc6 = nn.Conv2d(512, 1024, 3, padding=6, dilation=6)
        self.fc7 = nn.Conv2d(1024, 1024, 1)
        self.conv4_3_norm = L2Normalization(gamma_init=20, n_channels=512, name='conv4_3_norm')

    def _vgg_to_conv4_3(self, x):
        ca, cap = self.conv_act, self.conv_act_pool
        x = self.conv1_block_pool(self.conv1_1, self.conv1_2, x)                # conv1_1 -> conv1_2 -> pool1 ('same' pooling pads bottom/right)
        # (training step: each pair of consecutive ReLU layers shares a link -- the upper layer's data gradient leaves its kernel masked
        #  by the lower layer's activation, which is that layer's ReLU backward: models/_common.py, _ReluLink; None under no_grad)
        l2, l31, l32, l41, l42 = (self.relu_link() for _ in range(5))
        x = cap(self.conv2_2, ca(self.conv2_1, x, link_out=l2), 2, 2, ceil_mode=True, link_in=l2)
        x = cap(self.conv3_3, ca(self.conv3_2, ca(self.conv3_1, x, link_out=l31), link_in=l31, link_out=l32), 2, 2, ceil_mode=True, link_in=l32)
        return ca(self.conv4_3, ca(self.conv4_2, ca(self.conv4_1, x, link_out=l41), link_in=l41, link_out=l42), link_in=l42)

    def _vgg_from_pool4(self, x):
        ca, cap = self.conv_act, self.conv_act_pool
        x = cap(self.conv5_3, ca(self.conv5_2, ca(self.conv5_1, x)), 3, 1, pad=1)
        return ca(self.fc7, ca(self.fc6, x))

    def _vgg_from_conv4_3(self, conv4_3):
        return self._vgg_from_pool4(self.max_pool(conv4_3, 2, 2, ceil_mode=True))

    def _vgg(self, x):
        conv4_3 = self._vgg_to_conv4_3(x)
        return conv4_3, self._vgg_from_conv4_3(conv4_3)

    def _trunk(self, x):
        """[conv4_3_norm, fc7].  Fused bf16 inference (round 6): pool4 and conv4_3_norm are ONE pass over the conv4_3 map
        (csrc/ssdhip_layers.hip, pool2_l2norm_kernel: 15.9 + 21 -> ~24 us at batch 32, bit-identical to the two passes)."""
        import os
        conv4_3 = self._vgg_to_conv4_3(x)
        norm = self.conv4_3_norm
        if (self._fused(conv4_3) and conv4_3.shape[1] == 512 and norm.fused_inference and norm.gamma is not None
                and not torch.is_grad_enabled() and os.environ.get("SSDHIP_NO_POOL_NORM", "0") != "1"):
            from .. import _native as nat
            pooled, normed = nat.pool2_l2_normalize(conv4_3, norm.gamma_float32())
            return [normed, self._vgg_from_pool4(pooled)]
        return [norm(conv4_3), self._vgg_from_conv4_3(conv4_3)]

    @staticmethod
    def _vgg_sizes(n):
        for _ in range(3):
            n = pool_out(n, 2, 2, ceil_mode=True)
        c43 = n
        n = pool_out(n, 2, 2, ceil_mode=True)
        return c43, n                                                   # conv4_3, fc7


class SSD300(_VGGBase):
    SOURCE_CHANNELS = (512, 1024, 512, 256, 256, 256)
    NAMES = ('conv4_3_norm', 'fc7', 'conv6_2', 'conv7_2', 'conv8_2', 'conv9_2')

    def __init__(self, image_size, n_classes, mode, l2_regularization, scales, aspect_ratios, n_boxes, steps, offsets,
                 two_boxes_for_ar1, clip_boxes, variances, coords, normalize_coords, subtract_mean, divide_by_stddev,
                 swap_channels, confidence_thresh, iou_threshold, top_k, nms_max_output_size):
        super().__init__(image_size, n_classes, mode, l2_regularization, subtract_mean, divide_by_stddev, swap_channels,
                         confidence_thresh, iou_threshold, top_k, nms_max_output_size, coords, normalize_coords)
        self._build_vgg(self.img_channels)
        self.conv6_1, self.conv6_2 = nn.Conv2d(1024, 256, 1), nn.Conv2d(256, 512, 3, stride=2, padding=1)
        self.conv7_1, self.conv7_2 = nn.Conv2d(512, 128, 1), nn.Conv2d(128, 256, 3, stride=2, padding=1)
        self.conv8_1, self.conv8_2 = nn.Conv2d(256, 128, 1), nn.Conv2d(128, 256, 3)
        self.conv9_1, self.conv9_2 = nn.Conv2d(256, 128, 1), nn.Conv2d(128, 256, 3)
        self.conf_heads = nn.ModuleList([nn.Conv2d(ch, nb * self.n_classes, 3, padding=1)
                                         for ch, nb in zip(self.SOURCE_CHANNELS, n_boxes)])
        self.loc_heads = nn.ModuleList([nn.Conv2d(ch, nb * 4, 3, padding=1) for ch, nb in zip(self.SOURCE_CHANNELS, n_boxes)])
        self.priorboxes = make_priorboxes(self.img_height, self.img_width, scales, aspect_ratios, two_boxes_for_ar1, steps,
                                          offsets, clip_boxes, variances, coords, normalize_coords,
                                          [n + '_mbox_priorbox' for n in self.NAMES])
        he_normal_(self)

    def trunk_features(self, x):
        """The two source maps the VGG trunk yields (conv4_3 after L2Normalization, fc7): their predictor heads do not depend on
        the extra layers, which `extra_features` derives from fc7."""
        return self._trunk(x)

    def extra_features(self, fc7):
        return self.extra_features_tail(self.extra_features_front(fc7))

    def extra_features_front(self, fc7):
        """conv6_1 -> conv6_2: the part of the extra layers that still wants the whole chip (split-K launches)."""
        return self.conv_act(self.conv6_2, self.conv_act(self.conv6_1, fc7))

    def extra_features_tail(self, conv6_2):
        """conv7_1 ... conv9_2 from conv6_2; returns [conv6_2, conv7_2, conv8_2, conv9_2]."""
        ca = self.conv_act
        tail = self._extras_tail_chain(conv6_2)
        if tail is not None:
            return [conv6_2] + tail
        conv7_2 = ca(self.conv7_2, ca(self.conv7_1, conv6_2))
        conv8_2 = ca(self.conv8_2, ca(self.conv8_1, conv7_2))
        conv9_2 = ca(self.conv9_2, ca(self.conv9_1, conv8_2))
        return [conv6_2, conv7_2, conv8_2, conv9_2]

    def _extras_tail_chain(self, conv6_2, _refresh_only=False):
        """conv7_1 ... conv9_2 (reference models/keras_ssd300.py:304-313) as ONE launch, one workgroup per image, the intermediate maps in
        LDS (csrc/ssdhip_chain.hip) on the fused bf16 inference path; None -> the caller runs the six layers one by one.  The filters are
        re-packed in MFMA fragment order once per set of weights (keyed on the parameters' versions)."""
        import os
        if not _refresh_only and (not self._fused(conv6_2) or os.environ.get("SSDHIP_NO_CHAIN", "0") == "1"):
            return None
        convs = self._tail_convs()
        key = self._tail_chain_key() + (str(conv6_2.device),)
        st = self.__dict__.get("_tail_chain")
        if st is None or st["key"] != key:
            # the same geometry on the same device: re-packed IN PLACE (a captured HIP graph, GraphedInference, keeps reading these
            # storages); anything else builds fresh tensors
            old = st["layers"] if (st is not None and st["layers"] is not None and st["key"][-1] == key[-1]) else None
            layers = []
            with torch.no_grad():
                for i, c in enumerate(convs):
                    wb = c.weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                    if old is not None:
                        packed = nat.conv_chain_pack(wb, out=old[i]["packed"])
                        bias = old[i]["bias"]
                        bias.copy_(c.bias.detach())
                    else:
                        packed = nat.conv_chain_pack(wb)
                        bias = c.bias.detach().to(torch.bfloat16).clone()
                    if packed is None:
                        layers = None
                        break
                    layers.append(dict(packed=packed, bias=bias, k=c.kernel_size[0], stride=c.stride[0],
                                       pad=c.padding[0], cout=c.out_channels, relu=1, keep=bool(i & 1)))
            st = {"key": key, "layers": layers}
            self.__dict__["_tail_chain"] = st
        if st["layers"] is None or _refresh_only:
            return None
        return nat.conv_chain(conv6_2, st["layers"])

    def _tail_convs(self):
        return [self.conv7_1, self.conv7_2, self.conv8_1, self.conv8_2, self.conv9_1, self.conv9_2]

    def _tail_chain_key(self):
        return tuple((id(c.weight), c.weight.data_ptr(), c.weight._version, c.bias.data_ptr(), c.bias._version) for c in self._tail_convs())

    def _derived_weights_key(self):
        return super()._derived_weights_key() + self._tail_chain_key()

    def _refresh_derived_weights(self):
        """The fragment-packed conv7_1 ... conv9_2 filters beside the base class's caches, each re-built in its own storage."""
        super()._refresh_derived_weights()
        st = self.__dict__.get("_tail_chain")
        if st is not None and st["layers"] is not None and st["key"][:-1] != self._tail_chain_key():
            dev = st["layers"][0]["packed"].device
            self._extras_tail_chain(torch.empty((0,), dtype=torch.bfloat16, device=dev), _refresh_only=True)

    def features(self, x):
        early = self.trunk_features(x)
        return early + self.extra_features(early[1])

    def predictor_sizes(self):
        out = []
        for n in (self.img_height, self.img_width):
            c43, f7 = self._vgg_sizes(n)
            c6 = conv_out(f7, 3, 2, 1)
            c7 = conv_out(c6, 3, 2, 1)
            c8 = conv_out(c7, 3)
            c9 = conv_out(c8, 3)
            out.append([c43, f7, c6, c7, c8, c9])
        return np.array(list(zip(*out)))


def ssd_300(image_size, n_classes, mode='training', l2_regularization=0.0005, min_scale=None, max_scale=None, scales=None,
            aspect_ratios_global=None,
            aspect_ratios_per_layer=[[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0/3.0], [1.0, 2.0, 0.5, 3.0, 1.0/3.0],
                                     [1.0, 2.0, 0.5, 3.0, 1.0/3.0], [1.0, 2.0, 0.5], [1.0, 2.0, 0.5]],
            two_boxes_for_ar1=True, steps=[8, 16, 32, 64, 100, 300], offsets=None, clip_boxes=False,
            variances=[0.1, 0.1, 0.2, 0.2], coords='centroids', normalize_coords=True, subtract_mean=[123, 117, 104],
            divide_by_stddev=None, swap_channels=[2, 1, 0], confidence_thresh=0.01, iou_threshold=0.45, top_k=200,
            nms_max_output_size=400, return_predictor_sizes=False):
    '''Build an SSD300 (reference keras_ssd300.py:31-59 for the arguments).  Returns a torch module whose
    forward takes `(batch, height, width, channels)` images (0..255, RGB) and returns the
    `(batch, 8732, n_classes+1+12)` prediction tensor (`mode='training'`) or the decoded
    `(batch, top_k, 6)` detections (`'inference'`, `'inference_fast'`); optionally also `predictor_sizes`.'''
    scales, ars, n_boxes, steps, offsets = resolve_anchor_config(6, min_scale, max_scale, scales, aspect_ratios_global,
                                                                 aspect_ratios_per_layer, two_boxes_for_ar1, steps,
                                                                 offsets, variances)
    model = SSD300(image_size, n_classes, mode, l2_regularization, scales, ars, n_boxes, steps, offsets, two_boxes_for_ar1,
                   clip_boxes, variances, coords, normalize_coords, subtract_mean, divide_by_stddev, swap_channels,
                   confidence_thresh, iou_threshold, top_k, nms_max_output_size)
    if return_predictor_sizes:
        return model, model.predictor_sizes()
    return model
