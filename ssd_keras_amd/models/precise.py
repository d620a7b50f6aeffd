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

float16 range (VERDICT r3 weak #7 / ADVICE r3).  An activation is carried as hi = fl16(x), lo = fl16(x - hi): above 65 504 hi is inf and
the layer is poisoned; below ~6e-5 the lo part leaves float16's normal range and the 2^-22 precision degrades.  The reference's float32
graph has no such limit.  So every layer's map is stored DIVIDED by a per-layer power of two (`calibrate`: the float32 framework model run
once, layer by layer, on a representative batch; a map's largest magnitude is brought to <= 2^14, four times below the limit), folded
into the kernel's output scale and bias -- exact, powers of two -- and undone wherever float32 values leave the pair representation
(pooling inputs, L2Normalization, the predictor outputs).  The first call calibrates on its own batch; and every eager call checks that
the predictions are finite and raises FloatingPointError instead of returning inf / NaN (`check_finite=False` to skip the sync).

`model.precise()` (models/_common.py) makes this the model's forward path: `model(images)` then returns the float32 predictions (or the
decoded detections in the inference modes) computed here.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import _native as nat


class PreciseForward:
    LIMIT = 16384.0                                       # a calibrated map's largest magnitude after scaling (float16 max: 65 504)

    def __init__(self, model, stream_priority=0, check_finite=True):
        if not hasattr(model, "_vgg") or not hasattr(model, "extra_features"):
            raise TypeError("PreciseForward mirrors the VGG-based builders (ssd_300 / ssd_512)")
        if next(model.parameters()).dtype != torch.float32:
            raise TypeError("PreciseForward takes the float32 model (the reference's precision)")
        self.model = model
        self._packed = {}
        self._side = {}
        self._scale = {}                                  # id(conv) -> power-of-two divisor of the layer's stored output
        self._calibrating = None                          # dict while `calibrate` records the layers' largest magnitudes
        self._calibrated = False
        self._headroom = 4.0
        import os as _os
        self._framework_calibration = _os.environ.get("SSDHIP_X3_FRAMEWORK_CALIBRATION", "0") == "1"   # rounds 3-5: MIOpen float32 walk
        self.check_finite = check_finite
        self._last_heads = None
        import os
        self._torch_assembly = os.environ.get("SSDHIP_X3_TORCH_ASSEMBLY", "0") == "1"    # the round-4 assembly in framework ops (A/B)
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

        s_out = self._scale.get(id(conv), 1.0)

        def build():
            w, oscale = nat.x3_pack_weight(conv.weight, slab64=slab64)
            return w, oscale, (conv.bias.detach().float() / s_out).contiguous() if conv.bias is not None else None
        return self._pack((id(conv), s_out), [conv.weight] + ([conv.bias] if conv.bias is not None else []), build)

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

    # An activation travels as (tensor, s): the TRUE float32 map is tensor * s -- tensor is the float16 [hi | lo] pair map, or a float32
    # map where a layer leaves the pair representation (out_f32), or the true float32 map itself while calibrating (s = 1).
    def _true(self, act):
        t, sc = act
        if t.dtype == torch.float16:
            t = nat.x3_merge(t)
        return t * sc if sc != 1.0 else t

    def _as_pair(self, act):
        t, sc = act
        if t.dtype == torch.float16:
            return act
        return nat.x3_split(t.contiguous(memory_format=torch.channels_last)), sc

    PROBE = 2.0 ** 14                                     # calibration probe: the layer's output divided by this cannot leave float16

    def _divisor(self, amax):
        import math
        if not math.isfinite(amax) or amax <= 65504.0 / self._headroom:
            return 1.0
        return 2.0 ** math.ceil(math.log2(amax / self.LIMIT))

    def conv(self, conv, act, relu=True, pool=False, out_f32=False):
        if self._calibrating is not None and self._supported(conv) and act[0].is_cuda and not self._framework_calibration:
            # Round 6: the layer calibrates ITSELF.  Pass 1 runs the X3 kernel with the output divided by 2^14 (nothing a float32 network
            # on 0..255 images produces leaves float16 there; what underflows is far below the range question) and reads the largest
            # magnitude; the divisor is chosen from it as before; the normal path below is pass 2 and hands the next layer exactly the
            # map a calibrated forward would.  (Rounds 3-5 walked the float32 FRAMEWORK model layer by layer: MIOpen serves float32
            # NHWC convolutions from its naive kernel, 85-450 ms per layer -- the first call of model.precise() took 25 s.)
            x2, s_in = self._as_pair(act)
            w, oscale, _b = self._conv_filters(conv)
            import math
            probe = self.PROBE
            for _ in range(6):                            # (a probe that overflows all the same is repeated with its square: 2^28, 2^56, ...)
                b_try = (conv.bias.detach().float() / probe).contiguous() if conv.bias is not None else None
                y = nat.conv2d_x3(x2, w, b_try, oscale * s_in / probe, stride=conv.stride[0], padding=conv.padding[0],
                                  dilation=conv.dilation[0], relu=relu, pool=pool, out_f32=True)
                amax = float(y.abs().max()) * probe
                del y
                if math.isfinite(amax) or probe > 1e30:
                    break
                probe = probe * probe
            self._calibrating[id(conv)] = amax
            sc = self._divisor(amax)
            if sc != 1.0:
                self._scale[id(conv)] = sc
            else:
                self._scale.pop(id(conv), None)
        elif self._calibrating is not None:               # the float32 framework convolution on true values, magnitudes recorded
            y = F.conv2d(self._true(act), conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation)
            y = torch.relu(y) if relu else y
            if pool:
                y = F.max_pool2d(y, 2, 2, ceil_mode=True)
            self._calibrating[id(conv)] = float(y.abs().max())
            return y.contiguous(memory_format=torch.channels_last), 1.0
        s_out = self._scale.get(id(conv), 1.0)
        if not self._supported(conv):
            # a layer the kernel does not cover (SSD512's 4x4 conv10_2): the float32 framework convolution on the merged activation
            y = F.conv2d(self._true(act), conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation)
            y = torch.relu(y) if relu else y
            if pool:
                y = F.max_pool2d(y, 2, 2, ceil_mode=True)
            y = (y / s_out).contiguous(memory_format=torch.channels_last)
            return (y, s_out) if out_f32 else (nat.x3_split(y), s_out)
        x2, s_in = self._as_pair(act)
        w, oscale, b = self._conv_filters(conv)
        # stored output = act(true sum + bias) / s_out = act((oscale s_in / s_out) acc + bias / s_out): powers of two, exact
        y = nat.conv2d_x3(x2, w, b, oscale * s_in / s_out, stride=conv.stride[0], padding=conv.padding[0], dilation=conv.dilation[0],
                          relu=relu, pool=pool, out_f32=out_f32)
        return y, s_out

    @staticmethod
    def _pool(t, kernel, stride, padding, ceil_mode):
        """MaxPooling2D of an activation in either representation: pair maps through libssdhip, float32 maps (calibration, layers
        the kernels do not cover) through the framework."""
        if t.dtype == torch.float16 and (t.shape[1] // 2) % 8 == 0:
            return nat.x3_maxpool(t, kernel, stride, padding, ceil_mode)
        if t.dtype == torch.float16:
            t = nat.x3_merge(t)
            return nat.x3_split(F.max_pool2d(t, kernel, stride, padding, ceil_mode=ceil_mode).contiguous(memory_format=torch.channels_last))
        return F.max_pool2d(t, kernel, stride, padding, ceil_mode=ceil_mode)

    @staticmethod
    def _same3(conv):
        return conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)

    def _raw_images(self, images):
        """The generator's (B, H, W, 3) float32 batch itself when conv1_1's kernel can apply the input Lambdas while it stages its input
        (round 6: mean subtraction, channel swap and the layout copy were three framework passes, 52 us of the step), else None."""
        m = self.model
        c11 = m.conv1_1
        if (self._calibrating is not None or os.environ.get("SSDHIP_X3_NO_FUSED_INPUT", "0") == "1" or images.dim() != 4
                or images.shape[-1] != 3 or images.shape[1] == 3 or getattr(m, "img_channels", 3) != 3 or not images.is_cuda
                or images.dtype != torch.float32 or not images.is_contiguous()):
            return None
        if not (c11.in_channels == 3 and c11.out_channels == 64 and self._same3(c11) and self._scale.get(id(c11), 1.0) == 1.0
                and c11.weight.dtype == torch.float32):
            return None
        return images

    def _input_constants(self):
        m = self.model

        def three(v):
            if v is None:
                return None
            v = [float(t) for t in v] if hasattr(v, "__len__") else [float(v)] * 3
            return v if len(v) == 3 else None
        mean, div = three(m.subtract_mean), three(m.divide_by_stddev)
        if (m.subtract_mean is not None and mean is None) or (m.divide_by_stddev is not None and div is None):
            raise ValueError("per-channel constants must have three entries")
        swap = [int(t) for t in m.swap_channels] if m.swap_channels else None
        if swap is not None and len(swap) != 3:
            raise ValueError("swap_channels must have three entries here")
        return mean, div, swap

    def _vgg(self, x, raw=None):
        m = self.model
        c = self.conv
        # conv1_1: three input channels, K = 27 -- float32 vector arithmetic with the split written directly (ssdhip_conv1_1_x3_nhwc;
        # the framework's float32 convolution of a 3-channel NHWC image is MIOpen's naive kernel: 5.2 ms at batch 32)
        c11 = m.conv1_1
        if raw is not None:
            mean, div, swap = self._input_constants()
            a = (nat.conv1_1_x3_pre(raw, c11.weight, c11.bias, mean, div, swap, relu=True), 1.0)
        elif self._calibrating is not None:
            y = torch.relu(F.conv2d(x, c11.weight, c11.bias, c11.stride, c11.padding))
            self._calibrating[id(c11)] = float(y.abs().max())
            s11 = 1.0 if self._framework_calibration else self._divisor(self._calibrating[id(c11)])   # (the next layer's probe takes pairs)
            a = ((y / s11 if s11 != 1.0 else y).contiguous(memory_format=torch.channels_last), s11)
        elif (c11.in_channels == 3 and c11.out_channels == 64 and self._same3(c11) and x.is_cuda
                and x.permute(0, 2, 3, 1).is_contiguous() and self._scale.get(id(c11), 1.0) == 1.0):
            a = (nat.conv1_1_x3(x, c11.weight, c11.bias, relu=True), 1.0)
        else:                                             # (also: a first layer whose map needs a divisor -- the kernel has no scale)
            s11 = self._scale.get(id(c11), 1.0)
            a = (nat.x3_split((torch.relu(F.conv2d(x, c11.weight, c11.bias, c11.stride, c11.padding)) / s11)
                              .contiguous(memory_format=torch.channels_last)), s11)
        a = c(m.conv1_2, a, pool=True)                                     # MaxPooling2D(2, 2, 'same') fused (:275-276)
        a = c(m.conv2_2, c(m.conv2_1, a), pool=True)
        a = c(m.conv3_3, c(m.conv3_2, c(m.conv3_1, a)), pool=True)
        # round 6: conv4_3 / conv5_3 stay PAIR maps and pool4 / pool5 select pairs (ssdhip_x3_maxpool_nhwc) -- before: a float32 map
        # out of the convolution, the framework's float32 pooling, a split pass (0.25 ms of glue per step, twice the bytes)
        conv4_3 = c(m.conv4_3, c(m.conv4_2, c(m.conv4_1, a)))                        # (pair map / s, s)
        a = (self._pool(conv4_3[0], 2, 2, 0, True), conv4_3[1])                      # pooling commutes with the positive divisor
        conv5_3 = c(m.conv5_3, c(m.conv5_2, c(m.conv5_1, a)))
        a = (self._pool(conv5_3[0], 3, 1, 1, False), conv5_3[1])
        fc7 = c(m.fc7, c(m.fc6, a))
        return conv4_3, fc7

    def _extras_chain(self, act, convs):
        """The 1x1 / 3x3 pairs behind conv6_2 (conv7_1 ... conv9_2, reference models/keras_ssd300.py:304-313) as ONE launch with the
        intermediate pair maps in LDS; returns the (pair map, divisor) of every second layer, or None -> the caller runs them one by
        one (calibration, a layer the kernel does not cover, maps that do not fit)."""
        if self._calibrating is not None or not convs or os.environ.get("SSDHIP_X3_NO_CHAIN", "0") == "1":
            return None
        t, s_in = act
        if t.dtype != torch.float16 or not t.is_cuda:
            return None
        layers = []
        for i, conv in enumerate(convs):
            if not self._supported(conv) or conv.dilation != (1, 1) or conv.bias is None:
                return None
            w, oscale, bias = self._conv_filters(conv)
            packed = self._pack(("chain", id(conv)), [conv.weight], lambda w=w: nat.conv_chain_x3_pack(w))
            if packed is None:
                return None
            s_out = self._scale.get(id(conv), 1.0)
            layers.append({"packed": packed, "bias": bias, "k": conv.kernel_size[0], "stride": conv.stride[0], "pad": conv.padding[0],
                           "cout": conv.out_channels, "relu": 1, "mul": oscale * s_in / s_out, "keep": bool(i & 1)})
            s_in = s_out
        outs = nat.conv_chain_x3(t, layers)
        if outs is None:
            return None
        return [(o, self._scale.get(id(c), 1.0)) for o, c in zip(outs, convs[1::2])]

    _EXTRA_NAMES = [("conv6_1", "conv6_2"), ("conv7_1", "conv7_2"), ("conv8_1", "conv8_2"), ("conv9_1", "conv9_2"), ("conv10_1", "conv10_2")]

    def _head(self, l, act):
        """conf + loc predictors of source map l in one launch -> (B, h, w, Cpad) float32 NHWC view (true values: divisor 1)."""
        m = self.model
        ch, lh = m.conf_heads[l], m.loc_heads[l]
        if not (self._same3(ch) and self._same3(lh)):
            raise RuntimeError("predictor heads must be 3x3 'same' convolutions")
        if self._calibrating is not None and (self._framework_calibration or not act[0].is_cuda):
            x = self._true(act)
            y = torch.cat([F.conv2d(x, ch.weight, ch.bias, 1, 1), F.conv2d(x, lh.weight, lh.bias, 1, 1)], dim=1)
            return y.permute(0, 2, 3, 1)
        s2, s_in = self._as_pair(act)
        w, oscale, bias = self._head_filters(l)
        y = nat.conv2d_x3(s2, w, bias, oscale * s_in, stride=1, padding=1, dilation=1, relu=False, out_f32=True)   # (B, Cpad, h, w)
        return y.permute(0, 2, 3, 1)                      # NHWC view: the channel axis splits as (box, class) (:363-383)

    def _assemble(self, ys, decode):
        """The packed float32 head maps -> the prediction tensor (Reshape / Concatenate / softmax / AnchorBoxes, :363-419) in ONE
        libssdhip launch -- or, with `decode`, straight into the model's DecodeDetections layer without a prediction tensor in HBM
        (round 5; before: twelve reshaping copies, two concatenations, a softmax and an index_select, ~0.27 ms of the step)."""
        m = self.model
        maps = [y.permute(0, 3, 1, 2) for y in ys]        # (B, Cpad, h, w) channels_last: what the convolution wrote
        sizes = [(y.shape[1], y.shape[2]) for y in ys]
        anchors = m.anchors_and_variances(sizes, maps[0].device)
        n_boxes = [pb.n_boxes for pb in m.priorboxes]
        none = [None] * len(maps)
        if decode:
            return m.decoder.forward_from_heads(maps, none, none, none, n_boxes, anchors, m.n_classes)
        return nat.assemble_predictions(maps, none, none, none, n_boxes, anchors, m.n_classes)

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
    def calibrate(self, images, headroom=4.0):
        """Choose the per-layer power-of-two divisors from the layers' activations on `images`: a map whose largest magnitude exceeds
        65504 / headroom is stored divided by the power of two that brings it to <= LIMIT.  Each layer is probed by its own X3 kernel
        with a 2^14 divisor, then run for real (`conv`): two forwards' worth of kernels, no framework convolution (round 6; the float32
        framework walk of rounds 3-5 -- SSDHIP_X3_FRAMEWORK_CALIBRATION=1 -- took 25 s on MIOpen's naive NHWC kernels).  Returns
        {layer name: (largest magnitude, divisor)}."""
        import math
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("calibrate outside a stream capture")
        self._headroom = float(headroom)
        self._scale = {}
        self._calibrating = {}
        try:
            if str(images.device) not in self._side:
                # a placeholder (every head on the main stream) for the calibration pass only: removed whatever happens, so that a
                # failed calibration does not leave the measured stream choice switched off for good
                self._side[str(images.device)] = (torch.cuda.current_stream(images.device), torch.cuda.current_stream(images.device))
                try:
                    self._forward(images)
                finally:
                    del self._side[str(images.device)]
            else:
                self._forward(images)
            amax = self._calibrating
        finally:
            self._calibrating = None
        names = {id(mod): n for n, mod in self.model.named_modules() if isinstance(mod, nn.Conv2d)}
        self._scale, report = {}, {}
        for key, a in amax.items():
            if not math.isfinite(a):
                raise FloatingPointError("layer %s overflows float32 on the calibration batch" % names.get(key, key))
            sc = 1.0
            if a > 65504.0 / headroom:
                sc = 2.0 ** math.ceil(math.log2(a / self.LIMIT))
            if sc != 1.0:
                self._scale[key] = sc
            report[names.get(key, str(key))] = (a, sc)
        self._calibrated = True
        return report

    @torch.no_grad()
    def __call__(self, images, decode=False):
        """The float32 prediction tensor; `decode=True`: the model's DecodeDetections(Fast) output computed straight from the head
        maps (inference modes only)."""
        capturing = torch.cuda.is_current_stream_capturing()
        if not self._calibrated:
            if capturing:
                raise RuntimeError("call PreciseForward (or calibrate) once outside a stream capture first")
            self.calibrate(images)
        if str(images.device) not in self._side:
            if capturing:
                raise RuntimeError("call PreciseForward once outside a stream capture first (it times its side streams)")
            self._pick_streams(images)
        if decode and self.model.decoder is None:
            raise ValueError("decode=True needs a model built with mode='inference' / 'inference_fast' (this one has no decoder layer)")
        if decode and self.model.n_classes > 81:
            return self.model.decoder(self.__call__(images))
        y = self._forward(images, decode=decode)
        if self.check_finite and not capturing:
            # (decode: the range guard looks at the head maps -- a poisoned layer shows there as it would in the predictions)
            probe = self._last_heads if decode else [y]
            if not all(bool(torch.isfinite(t).all()) for t in probe):
                self._last_heads = None
                raise FloatingPointError("PreciseForward: non-finite predictions -- an activation left the float16 pair range (|x| >= "
                                         "65504 after the calibrated per-layer divisors); re-run calibrate() on a batch like this one")
        self._last_heads = None                                   # (the float32 head maps are not kept alive between calls)
        return y

    @torch.no_grad()
    def _forward(self, images, decode=False):
        m = self.model
        raw = self._raw_images(images)
        x = m.preprocess(images) if raw is None else None                     # float32, channels_last
        conv4_3, fc7 = self._vgg(x, raw)
        n_heads = len(m.conf_heads)
        names = [(a, b) for a, b in self._EXTRA_NAMES if hasattr(m, a)]
        if 2 + len(names) != n_heads:
            raise RuntimeError("this builder's extra layers are not the ones PreciseForward knows")
        # Three streams, as in the bf16 step (models/_common.py): the extra layers are a chain of eight small launches that leaves
        # most CUs idle (0.44 ms), the two trunk heads do not depend on it (0.42 ms), and each small head only needs its own source
        # map.  Main: the chain.  Side 1: conv4_3_norm + its head, fc7's head.  Side 2: the head of every extra map as soon as it
        # exists.  One after the other they took 1.2 ms of an 8.3 ms forward (profiles/r03ze_x3_timeline.json).
        main = torch.cuda.current_stream(images.device)
        side1, side2 = self._streams(images.device)
        one_stream = side1.cuda_stream == main.cuda_stream          # calibration: everything on the current stream
        trunk = torch.cuda.Event()
        trunk.record(main)
        ys = [None] * n_heads
        with torch.cuda.stream(side1):
            side1.wait_event(trunk)
            fc7_first = os.environ.get("SSDHIP_X3_FC7_HEAD_FIRST", "1") == "1"
            if fc7_first:
                ys[1] = self._head(1, fc7)
            nl = m.conv4_3_norm
            if nl.gamma is None:
                nl.build(conv4_3[0].shape[1] // 2 if conv4_3[0].dtype == torch.float16 else conv4_3[0].shape[1], images.device)
            if (conv4_3[0].dtype == torch.float16 and conv4_3[0].is_cuda and nl.gamma.dtype == torch.float32 and nl.gamma.is_contiguous()
                    and os.environ.get("SSDHIP_X3_NO_PAIR_NORM", "0") != "1"):
                # round 6: L2Normalization on the pair map itself (one pass; before: merge -> float32 normalisation -> split, 111 us
                # in front of the conv4_3 head on this stream)
                ys[0] = self._head(0, (nat.x3_l2_normalize(conv4_3[0], nl.gamma, conv4_3[1]), 1.0))
            else:
                norm = nl(self._true(conv4_3))                                   # L2Normalization on the true float32 map (:316)
                ys[0] = self._head(0, (norm.contiguous(memory_format=torch.channels_last), 1.0))
            if not fc7_first:
                ys[1] = self._head(1, fc7)
        if not one_stream:
            conv4_3[0].record_stream(side1)
            fc7[0].record_stream(side1)
        def small_head(k, act, on_main=False):
            if on_main:                                   # (behind the one-launch tail the main stream has nothing else to do)
                ys[2 + k] = self._head(2 + k, act)
                return
            ready = torch.cuda.Event()
            ready.record(main)
            # (round 6: the small heads ALTERNATE between the two side streams -- on one stream the 3 x 3 and 1 x 1 maps' heads, ~75 us of
            #  weight streaming each, queued behind one another at the very end of the step)
            hs = side1 if (k & 1) and os.environ.get("SSDHIP_X3_HEADS_ALTERNATE", "1") == "1" else side2
            with torch.cuda.stream(hs):
                hs.wait_event(ready)
                ys[2 + k] = self._head(2 + k, act)
            if not one_stream:
                act[0].record_stream(hs)

        x2 = fc7
        for k, (a, b) in enumerate(names):
            if k == 1:
                # round 6: conv7_1 ... conv9_2 in ONE launch (csrc/ssdhip_chain.hip, conv_chain_x3_kernel; six launches of 22-102 us before)
                tail = self._extras_chain(x2, [getattr(m, n) for pair in names[1:] for n in pair])
                if tail is not None:
                    # the tail's three maps exist at once: their heads on the main stream and on side stream 2 -- side stream 1 is busy
                    # with the two trunk heads until later than that (r06zk timeline)
                    on_main = os.environ.get("SSDHIP_X3_TAIL_HEADS_ON_MAIN", "1") == "1"
                    for kk, act in enumerate(tail, start=1):
                        if not (on_main and (kk & 1)):
                            small_head(kk, act)
                    for kk, act in enumerate(tail, start=1):
                        if on_main and (kk & 1):
                            small_head(kk, act, on_main=True)
                    break
            x2 = self.conv(getattr(m, b), self.conv(getattr(m, a), x2))       # main stream
            small_head(k, x2)
        if not one_stream:
            main.wait_stream(side1)
            main.wait_stream(side2)
        b = images.shape[0]
        if self._calibrating is None and all(y.dtype == torch.float32 and y.is_cuda for y in ys) and not self._torch_assembly:
            if not one_stream:
                for y in ys:
                    y.record_stream(main)
            self._last_heads = ys if decode else None
            return self._assemble(ys, decode)
        confs, locs, sizes = [], [], []
        for l, y in enumerate(ys):
            if not one_stream:
                y.record_stream(main)
            ch, lh = m.conf_heads[l], m.loc_heads[l]
            confs.append(y[..., :ch.out_channels].reshape(b, -1, m.n_classes))
            locs.append(y[..., ch.out_channels:ch.out_channels + lh.out_channels].reshape(b, -1, 4))
            sizes.append((y.shape[1], y.shape[2]))
        conf = torch.softmax(torch.cat(confs, dim=1), dim=-1)                  # 'mbox_conf_softmax' (:415)
        loc = torch.cat(locs, dim=1)
        anchors = m.anchors_and_variances(sizes, conf.device)
        pred = torch.cat([conf, loc, anchors.unsqueeze(0).expand(b, -1, -1)], dim=2)   # 'predictions' (:419)
        self._last_heads = [pred] if decode else None
        return m.decoder(pred) if decode else pred
