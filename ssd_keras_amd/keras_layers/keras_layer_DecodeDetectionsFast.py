"""`DecodeDetectionsFast` -- drop-in for keras_layers/keras_layer_DecodeDetectionsFast.py:29-266.

Class = argmax over all scores (:126-128), background dropped (:174), strict `>` confidence threshold
(:180), one class-agnostic NMS capped at `nms_max_output_size` (:199), top-k + zero padding.
Same kernels as `DecodeDetections` with `class_agnostic = 1`.
"""
from __future__ import annotations

from .keras_layer_DecodeDetections import DecodeDetections


class DecodeDetectionsFast(DecodeDetections):
    _class_agnostic = True
