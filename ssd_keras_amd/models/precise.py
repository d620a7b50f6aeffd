"""The SSD300 / SSD512 forward pass at the REFERENCE's precision on the MFMA path.

The reference's graph is float32 end to end (models/keras_ssd300.py:274-419).  The benchmarked backbone is bf16; MIOpen's float32
convolutions are bound by the 157 TFLOP/s float32 matrix rate.  `PreciseForward` runs the same float32 model through
`ssdhip_conv2d_x3_nhwc_f16` (csrc/ssdhip_conv.hip, X3): every activation and filter is carried as a float16 (hi, lo) pair -- a float32
value to 2^-22 -- and a convolution is hi.hi + hi.lo + lo.hi in one float16 MFMA K loop with float32 accumulation, float32 bias and
activation, and a re-split in the epilogue.  conv1_1 (three input channels, K = 27) is float32 vector arithmetic in its own kernel; the
graph glue that is not a convolution (the Lambda input pipeline, pool4 / pool5, L2Normalization, Reshape / Concatenate / softmax /
AnchorBoxes) stays float32 PyTorch, with one-pass split / merge kernels between the two representations.

    model = ssd_300(...).cuda().to(memory_format=torch.channels_last).eval()        # float32 weights
    y_pred = PreciseForward(model)(images)                                          # (B, 8732, n_classes + 12) float32

Filters are re-packed when a parameter changes (version / storage check per call).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .. import _native as nat


class PreciseForward:
    def __init__(self, model, stream_priority=0):
        if not hasattr(model, "_vgg") or not hasattr(model, "extra_features"):
            raise TypeError("PreciseForward mirrors the VGG-based builders (ssd_300 / ssd_512)")
        if next(model.parameters()).dtype != torch.float32:
            raise TypeError("PreciseForward takes the float32 model (the reference's precision)")
        self.model = model
        self._packed = {}
        self._side = {}
        # The two side streams of __call__ are PICKED: which hardware queues a process's streams share depends on how many were created
        # before, and a pair that shares one with the default stream (or with each other) loses the overlap -- 6.9 ms against 7.4 ms
        # with ordinary streams, 6.9 against 8.6 - 9.0 ms with high-priority ones, 8.3 ms inside bench.py against 7.1 ms alone
        # (tools/time_x3_streams.py, r03zo / r03zz).  The first call per device times a few fresh pairs and keeps the fastest.
        self._stream_priority = stream_priority

    # -- filters ------------------------------------------------------------------------------------------------------------------
    def _pack(self, key, tensors, build):
        sig = tuple((id(t), t.data_ptr(), t._version) for t in tensors)
        hit = self._packed.get(key)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                hit = (sig, build())
            self._packed[key] = hit
        return hit[1]

    def _conv_filters(self, conv):
        # conv2_1 (64 -> 128 channels) goes to the slab kernel with a padded K loop (4/3 of the products, still faster than the
        # implicit-GEMM form: 523 us there, r03ze)
        slab64 = conv.in_channels == 64 and conv.out_channels % 128 == 0 and self._same3(conv)

        def build():
            w, oscale = nat.x3_pack_weight(conv.weight, slab64=slab64)
            return w, oscale, conv.bias.detach().float().contiguous() if conv.bias is not None else None
        return self._pack(id(conv), [conv.weight] + ([conv.bias] if conv.bias is not None else []), build)

    def _head_filters(self, l):
        """conf and loc filters of predictor layer l packed along Cout, zero rows up to a multiple of 128 (one launch per source map)."""
        ch, lh = self.model.conf_heads[l], self.model.loc_heads[l]

        def build():
            n = ch.out_channels + lh.out_channels
            pad = (-n) % 128                                  # a multiple of 128: the slab kernel's channel tile
            w = torch.cat([ch.weight, lh.weight] + ([ch.weight.new_zeros((pad,) + tuple(ch.weight.shape[1:]))] if pad else []), dim=0)
            b = torch.cat([ch.bias, lh.bias] + ([ch.bias.new_zeros((pad,))] if pad else []), dim=0).float().contiguous()
            pw, oscale = nat.x3_pack_weight(w)
            return pw, oscale, b
        return self._pack(("head", l), [ch.weight, lh.weight, ch.bias, lh.bias], build)

    # -- layers -------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _supported(conv):
        k = conv.kernel_size[0]
        return (k in (1, 3) and conv.kernel_size[1] == k and conv.stride[0] == conv.stride[1] and 1 <= conv.stride[0] <= 4
                and conv.groups == 1 and conv.dilation[0] == conv.dilation[1] and isinstance(conv.padding, tuple)
                and conv.padding[0] == conv.padding[1] and 0 <= conv.padding[0] <= conv.dilation[0] * (k // 2)
                and conv.padding_mode == 'zeros' and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0)

    def conv(self, conv, x2, relu=True, pool=False, out_f32=False):
        if not self._supported(conv):
            # a layer the kernel does not cover (SSD512's 4x4 conv10_2): the float32 framework convolution on the merged activation
            c = x2.shape[1] // 2
            y = F.conv2d(x2[:, :c].float() + x2[:, c:].float(), conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation)
            y = torch.relu(y) if relu else y
            if pool:
                y = F.max_pool2d(y, 2, 2, ceil_mode=True)
            y = y.contiguous(memory_format=torch.channels_last)
            return y if out_f32 else nat.x3_split(y)
        w, oscale, b = self._conv_filters(conv)
        return nat.conv2d_x3(x2, w, b, oscale, stride=conv.stride[0], padding=conv.padding[0], dilation=conv.dilation[0], relu=relu,
                             pool=pool, out_f32=out_f32)

    @staticmethod
    def _same3(conv):
        return conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)

    def _vgg(self, x):
        m = self.model
        c = self.conv
        # conv1_1: three input channels, K = 27 -- float32 vector arithmetic with the split written directly (ssdhip_conv1_1_x3_nhwc;
        # the framework's float32 convolution of a 3-channel NHWC image is MIOpen's naive kernel: 5.2 ms at batch 32)
        c11 = m.conv1_1
        if (c11.in_channels == 3 and c11.out_channels == 64 and self._same3(c11) and x.is_cuda
                and x.permute(0, 2, 3, 1).is_contiguous()):
            x2 = nat.conv1_1_x3(x, c11.weight, c11.bias, relu=True)
        else:
            x2 = nat.x3_split(torch.relu(F.conv2d(x, c11.weight, c11.bias, c11.stride, c11.padding)))
        x2 = c(m.conv1_2, x2, pool=True)                                   # MaxPooling2D(2, 2, 'same') fused (:275-276)
        x2 = c(m.conv2_2, c(m.conv2_1, x2), pool=True)
        x2 = c(m.conv3_3, c(m.conv3_2, c(m.conv3_1, x2)), pool=True)
        conv4_3 = c(m.conv4_3, c(m.conv4_2, c(m.conv4_1, x2)), out_f32=True)
        x2 = nat.x3_split(F.max_pool2d(conv4_3, 2, 2, ceil_mode=True))
        conv5_3 = c(m.conv5_3, c(m.conv5_2, c(m.conv5_1, x2)), out_f32=True)
        x2 = nat.x3_split(F.max_pool2d(conv5_3, 3, 1, 1))
        fc7 = c(m.fc7, c(m.fc6, x2))
        return conv4_3, fc7

    _EXTRA_NAMES = [("conv6_1", "conv6_2"), ("conv7_1", "conv7_2"), ("conv8_1", "conv8_2"), ("conv9_1", "conv9_2"), ("conv10_1", "conv10_2")]

    def _head(self, l, s2):
        """conf + loc predictors of source map l in one launch -> (B, h, w, Cpad) float32 NHWC view."""
        m = self.model
        ch, lh = m.conf_heads[l], m.loc_heads[l]
        if not (self._same3(ch) and self._same3(lh)):
            raise RuntimeError("predictor heads must be 3x3 'same' convolutions")
        w, oscale, bias = self._head_filters(l)
        y = nat.conv2d_x3(s2, w, bias, oscale, stride=1, padding=1, dilation=1, relu=False, out_f32=True)   # (B, Cpad, h, w)
        return y.permute(0, 2, 3, 1)                      # NHWC view: the channel axis splits as (box, class) (:363-383)

    def _streams(self, device):
        return self._side[str(device)]

    def _pick_streams(self, images, n_pairs=4):
        key = str(images.device)
        best = None
        for _ in range(n_pairs):
            pair = (torch.cuda.Stream(device=images.device, priority=self._stream_priority),
                    torch.cuda.Stream(device=images.device, priority=self._stream_priority))
            self._side[key] = pair
            self._forward(images)                                             # filters packed, allocator warm
            torch.cuda.synchronize(images.device)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._forward(images)
            self._forward(images)
            b.record()
            b.synchronize()
            t = a.elapsed_time(b)
            if best is None or t < best[0]:
                best = (t, pair)
        self._side[key] = best[1]

    @torch.no_grad()
    def __call__(self, images):
        if str(images.device) not in self._side:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("call PreciseForward once outside a stream capture first (it times its side streams)")
            self._pick_streams(images)
        return self._forward(images)

    @torch.no_grad()
    def _forward(self, images):
        m = self.model
        x = m.preprocess(images)                                              # float32, channels_last
        conv4_3, fc7 = self._vgg(x)
        n_heads = len(m.conf_heads)
        names = [(a, b) for a, b in self._EXTRA_NAMES if hasattr(m, a)]
        if 2 + len(names) != n_heads:
            raise RuntimeError("this builder's extra layers are not the ones PreciseForward knows")
        # Three streams, as in the bf16 step (models/_common.py): the extra layers are a chain of eight small launches that leaves
        # most CUs idle (0.44 ms), the two trunk heads do not depend on it (0.42 ms), and each small head only needs its own source
        # map.  Main: the chain.  Side 1: conv4_3_norm + its head, fc7's head.  Side 2: the head of every extra map as soon as it
        # exists.  One after the other they took 1.2 ms of an 8.3 ms forward (profiles/r03ze_x3_timeline.json).
        main = torch.cuda.current_stream(x.device)
        side1, side2 = self._streams(x.device)
        trunk = torch.cuda.Event()
        trunk.record(main)
        ys = [None] * n_heads
        with torch.cuda.stream(side1):
            side1.wait_event(trunk)
            norm = m.conv4_3_norm(conv4_3)                                    # L2Normalization on float32 (:316)
            ys[0] = self._head(0, nat.x3_split(norm.contiguous(memory_format=torch.channels_last)))
            ys[1] = self._head(1, fc7)
        conv4_3.record_stream(side1)
        fc7.record_stream(side1)
        x2 = fc7
        for k, (a, b) in enumerate(names):
            x2 = self.conv(getattr(m, b), self.conv(getattr(m, a), x2))       # main stream
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(side2):
                side2.wait_event(ready)
                ys[2 + k] = self._head(2 + k, x2)
            x2.record_stream(side2)
        main.wait_stream(side1)
        main.wait_stream(side2)
        b = x.shape[0]
        confs, locs, sizes = [], [], []
        for l, y in enumerate(ys):
            y.record_stream(main)
            ch, lh = m.conf_heads[l], m.loc_heads[l]
            confs.append(y[..., :ch.out_channels].reshape(b, -1, m.n_classes))
            locs.append(y[..., ch.out_channels:ch.out_channels + lh.out_channels].reshape(b, -1, 4))
            sizes.append((y.shape[1], y.shape[2]))
        conf = torch.softmax(torch.cat(confs, dim=1), dim=-1)                  # 'mbox_conf_softmax' (:415)
        loc = torch.cat(locs, dim=1)
        anchors = m.anchors_and_variances(sizes, conf.device)
        return torch.cat([conf, loc, anchors.unsqueeze(0).expand(b, -1, -1)], dim=2)   # 'predictions' (:419)
