"""`DecodeDetections` -- drop-in for keras_layers/keras_layer_DecodeDetections.py:27-283 as a torch module.

Input  `(batch, n_boxes_total, n_classes + 4 + 8)` float32 on the GPU,
output `(batch, top_k, 6)` float32 `[class_id, confidence, xmin, ymin, xmax, ymax]`, sorted by
confidence, zero padded -- the layer's container.  The whole body (decode :124-149, per-class
threshold :180, `tf.image.non_max_suppression(max_output_size=nms_max_output_size)` :195-199,
`tf.nn.top_k(sorted=True)` + padding :238-251) is ONE call into libssdhip.so
(`ssdhip_decode_detections`, semantics SSDHIP_SEM_KERAS): three kernels on the current stream,
no host sync, graph-capturable.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _native as nat


class DecodeDetections(nn.Module):
    _class_agnostic = False

    def __init__(self, confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400,
                 coords='centroids', normalize_coords=True, img_height=None, img_width=None, **kwargs):
        super().__init__()
        if normalize_coords and ((img_height is None) or (img_width is None)):
            raise ValueError("If relative box coordinates are supposed to be converted to absolute coordinates, the "
                             "decoder needs the image size in order to decode the predictions, but `img_height == {}` "
                             "and `img_width == {}`".format(img_height, img_width))
        if coords != 'centroids':
            raise ValueError("The DetectionOutput layer currently only supports the 'centroids' coordinate format.")
        self.confidence_thresh = confidence_thresh
        self.iou_threshold = iou_threshold
        self.top_k = top_k
        self.normalize_coords = normalize_coords
        self.img_height, self.img_width = img_height, img_width
        self.coords = coords
        self.nms_max_output_size = nms_max_output_size
        self.name = kwargs.get('name')
        self.timing_events = None          # a list: every call appends its (start, end) torch.cuda.Event pair (bench.py)

    def build(self, input_shape):
        """Keras' shape hook (reference :105-107): nothing to create here -- the layer has no weights."""
        self.input_shape_ = tuple(input_shape) if input_shape is not None else None

    @torch.no_grad()
    def forward(self, y_pred):
        y = y_pred.detach()
        if y.dtype != torch.float32:
            y = y.float()
        out, _, _ = nat.decode(y.contiguous(), self.confidence_thresh, self.iou_threshold, self.top_k,
                               self.nms_max_output_size, self._class_agnostic, nat.SEM_KERAS, 'centroids',
                               self.normalize_coords, self.img_height, self.img_width, 'half', nat.F32, self.top_k)
        return out

    call = forward

    @torch.no_grad()
    def forward_from_heads(self, confs, locs, conf_biases, loc_biases, n_boxes, anchors_var, n_classes):
        """The same layer fed by the predictor heads' bf16 outputs instead of the assembled `(batch, #boxes, #classes + 12)`
        tensor: bias, softmax, anchors and the decode happen in one kernel (`scan_heads_kernel`), the
        prediction tensor never exists in HBM.  Same result, bit for bit, as `forward(assembled tensor)`."""
        ev = None
        if self.timing_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        out, _, _ = nat.decode_from_heads(confs, locs, conf_biases, loc_biases, n_boxes, anchors_var, n_classes,
                                          self.confidence_thresh, self.iou_threshold, self.top_k, self.nms_max_output_size,
                                          self._class_agnostic, nat.SEM_KERAS, 'centroids', self.normalize_coords, self.img_height,
                                          self.img_width, 'half', nat.F32, self.top_k)
        if ev is not None:
            ev[1].record()
            self.timing_events.append(ev)
        return out

    def compute_output_shape(self, input_shape):
        batch_size, n_boxes, last_axis = input_shape
        return (batch_size, self.top_k, 6)

    def get_config(self):
        return {'confidence_thresh': self.confidence_thresh, 'iou_threshold': self.iou_threshold, 'top_k': self.top_k,
                'nms_max_output_size': self.nms_max_output_size, 'coords': self.coords,
                'normalize_coords': self.normalize_coords, 'img_height': self.img_height, 'img_width': self.img_width}
