"""The reference's Python call surface, read from SOURCE with `ast` -- the TensorFlow / Keras / OpenCV modules
cannot be imported here, their `def` lines can be parsed.  `extract(root, ...)` returns {module: {qualified name: [(parameter,
default source or None), ...]}}; tests/golden/make_golden.py stores the reference's, tests/test_api_surface.py compares the package's."""
import ast
import os

# module (path relative to the reference root / to ssd_keras_amd) -> the callables that make up the drop-in surface
SURFACE = {
    "bounding_box_utils/bounding_box_utils.py": ["convert_coordinates", "convert_coordinates2", "intersection_area", "intersection_area_", "iou"],
    "ssd_encoder_decoder/matching_utils.py": ["match_bipartite_greedy", "match_multi"],
    "ssd_encoder_decoder/ssd_input_encoder.py": ["SSDInputEncoder.__init__", "SSDInputEncoder.__call__",
                                                 "SSDInputEncoder.generate_anchor_boxes_for_layer", "SSDInputEncoder.generate_encoding_template"],
    "ssd_encoder_decoder/ssd_output_decoder.py": ["greedy_nms", "_greedy_nms", "_greedy_nms2", "decode_detections", "decode_detections_fast",
                                                  "decode_detections_debug", "_greedy_nms_debug", "get_num_boxes_per_pred_layer", "get_pred_layers"],
    "keras_loss_function/keras_ssd_loss.py": ["SSDLoss.__init__", "SSDLoss.smooth_L1_loss", "SSDLoss.log_loss", "SSDLoss.compute_loss"],
    "keras_layers/keras_layer_AnchorBoxes.py": ["AnchorBoxes.__init__"],
    "keras_layers/keras_layer_DecodeDetections.py": ["DecodeDetections.__init__"],
    "keras_layers/keras_layer_DecodeDetectionsFast.py": ["DecodeDetectionsFast.__init__"],
    "keras_layers/keras_layer_L2Normalization.py": ["L2Normalization.__init__"],
    "models/keras_ssd300.py": ["ssd_300"],
    "models/keras_ssd512.py": ["ssd_512"],
    "models/keras_ssd7.py": ["build_model"],
    "eval_utils/average_precision_evaluator.py": ["Evaluator.__init__", "Evaluator.__call__", "Evaluator.match_predictions",
                                                  "Evaluator.compute_precision_recall", "Evaluator.compute_average_precisions",
                                                  "Evaluator.compute_mean_average_precision"],
    "eval_utils/coco_utils.py": ["get_coco_category_maps", "predict_all_to_json"],
    "data_generator/object_detection_2d_misc_utils.py": ["apply_inverse_transforms"],
    "data_generator/object_detection_2d_image_boxes_validation_utils.py": ["BoundGenerator.__init__", "BoxFilter.__init__", "BoxFilter.__call__",
                                                                           "ImageValidator.__init__", "ImageValidator.__call__"],
    "data_generator/object_detection_2d_patch_sampling_ops.py": ["PatchCoordinateGenerator.__init__", "CropPad.__init__", "CropPad.__call__",
                                                                 "Crop.__init__", "Pad.__init__", "RandomPatch.__init__", "RandomPatch.__call__",
                                                                 "RandomPatchInf.__init__", "RandomPatchInf.__call__",
                                                                 "RandomMaxCropFixedAR.__init__", "RandomPadFixedAR.__init__"],
    "data_generator/data_augmentation_chain_original_ssd.py": ["SSDRandomCrop.__init__", "SSDRandomCrop.__call__", "SSDExpand.__init__",
                                                               "SSDExpand.__call__", "SSDPhotometricDistortions.__init__",
                                                               "SSDPhotometricDistortions.__call__", "SSDDataAugmentation.__init__",
                                                               "SSDDataAugmentation.__call__"],
}


def _src(node, consts):
    """Source of a default value; a bare name bound to a literal at module level (`_DEFAULT_FORMAT`) is written out."""
    if node is None:
        return None
    if isinstance(node, ast.Name) and node.id in consts:
        node = consts[node.id]
    return ast.unparse(node)


def _params(fn, consts):
    a = fn.args
    pos = list(a.posonlyargs) + list(a.args)
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    out = [(p.arg, _src(d, consts)) for p, d in zip(pos, defaults)]
    if a.vararg:
        out.append(("*" + a.vararg.arg, None))
    for p, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append((p.arg, _src(d, consts)))
    if a.kwarg:
        out.append(("**" + a.kwarg.arg, None))
    return out


def _find(tree, qual):
    """The def of `qual` ('f' or 'Class.method'); a method missing from a class is looked up in its base classes of the same module."""
    parts = qual.split(".")
    if len(parts) == 1:
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name == parts[0]:
                return node
        return None
    classes = {n.name: n for n in tree.body if isinstance(n, ast.ClassDef)}
    seen, todo = set(), [parts[0]]
    while todo:
        c = todo.pop(0)
        if c in seen or c not in classes:
            continue
        seen.add(c)
        for node in classes[c].body:
            if isinstance(node, ast.FunctionDef) and node.name == parts[1]:
                return node
        todo += [b.id for b in classes[c].bases if isinstance(b, ast.Name)]
    return None


def extract(root, surface=None):
    out = {}
    for rel, names in (surface or SURFACE).items():
        path = os.path.join(root, rel)
        tree = ast.parse(open(path).read())
        consts = {}
        for node in tree.body:
            if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
                try:
                    ast.literal_eval(node.value)
                    consts[node.targets[0].id] = node.value
                except (ValueError, SyntaxError):
                    pass
        out[rel] = {}
        for q in names:
            node = _find(tree, q)
            out[rel][q] = _params(node, consts) if node is not None else None
    return out
