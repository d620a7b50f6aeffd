"""Drop-in for the reference's eval_utils/coco_utils.py: `get_coco_category_maps` :29-60 and `predict_all_to_json` :62-200 -- one
of the two callers of `decode_detections` the survey names (the other is `Evaluator.predict_on_dataset`).

`predict_all_to_json` walks a dataset batch by batch, predicts, decodes when the model was built in 'training' mode (here on the
GPU: `ssd_output_decoder.decode_detections` -> libssdhip), maps the boxes back to the original images with the transformations'
inverters and writes the MS COCO detection results JSON.  Same signature, keyword names, error text and output file as the reference;
`data_generator` is any object honouring the reference generator's `generate(...)` / `get_dataset_size()` contract
(data_generator/object_detection_2d_data_generator.py:873-1160 -- the generator itself is outside this package),
and it receives the SAME transformation list the reference builds (ConvertTo3Channels [, RandomPadFixedAR], Resize: all three are
mirrored here).  `model.predict(batch_X)` may return a NumPy array or a tensor (a torch module's `__call__` is used when it has no
`predict`)."""
from __future__ import annotations

import json
import sys
from math import ceil

import numpy as np

from ..data_generator.object_detection_2d_geometric_ops import Resize
from ..data_generator.object_detection_2d_misc_utils import apply_inverse_transforms
from ..data_generator.object_detection_2d_patch_sampling_ops import RandomPadFixedAR
from ..data_generator.object_detection_2d_photometric_ops import ConvertTo3Channels


def get_coco_category_maps(annotations_file):
    '''The maps between the 80 non-consecutive MS COCO category ids ('cats', spread over 1..90) and the consecutive ids a one-hot
    classifier needs ('classes', background = 0), in the order the annotation file lists its categories (reference :29-60).

    Returns `cats_to_classes`, `classes_to_cats`, `cats_to_names` (dicts) and `classes_to_names` (a list whose index is the class id).'''
    with open(annotations_file, 'r') as f:
        annotations = json.load(f)
    cats_to_classes, classes_to_cats, cats_to_names = {}, {}, {}
    classes_to_names = ['background']                     # index 0, so that the list index IS the class id
    for i, cat in enumerate(annotations['categories']):
        cats_to_classes[cat['id']] = i + 1
        classes_to_cats[i + 1] = cat['id']
        cats_to_names[cat['id']] = cat['name']
        classes_to_names.append(cat['name'])
    return cats_to_classes, classes_to_cats, cats_to_names, classes_to_names


def _predict(model, batch_X):
    """`model.predict(batch_X)` as Keras has it; a torch module is called under `no_grad` on a float32 copy of the batch on its device."""
    if hasattr(model, 'predict'):
        return model.predict(batch_X)
    import torch
    with torch.no_grad():
        if not torch.is_tensor(batch_X):
            try:
                device = next(model.parameters()).device
            except (StopIteration, AttributeError):
                device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
            batch_X = torch.as_tensor(np.asarray(batch_X), dtype=torch.float32, device=device)
        return model(batch_X)


def predict_all_to_json(out_file, model, img_height, img_width, classes_to_cats, data_generator, batch_size,
                        data_generator_mode='resize', model_mode='training', confidence_thresh=0.01, iou_threshold=0.45, top_k=200,
                        pred_coords='centroids', normalize_coords=True):
    '''Detection results of `model` over the whole of `data_generator`'s dataset as an MS COCO results file (reference :62-200).

    `data_generator_mode`: 'resize' warps every image to `(img_height, img_width)`, 'pad' first pads it to that aspect ratio.
    `model_mode`: 'training' -> the raw `(batch, #boxes, #classes + 12)` predictions are decoded here with `confidence_thresh`,
    `iou_threshold`, `top_k`, `pred_coords`, `normalize_coords`; 'inference' / 'inference_fast' -> the model's decoding layer already
    did, and only the all-zero padding rows are dropped.  Per box: `category_id = classes_to_cats[class id]`, the confidence rounded
    to three decimals, `bbox = [xmin, ymin, width, height]` from the corners rounded to one decimal.  Returns None.'''
    convert_to_3_channels = ConvertTo3Channels()
    resize = Resize(height=img_height, width=img_width)
    if data_generator_mode == 'resize':
        transformations = [convert_to_3_channels, resize]
    elif data_generator_mode == 'pad':
        # (the reference passes `clip_boxes=False` here, :125 -- not a parameter of RandomPadFixedAR, so its 'pad' mode raises
        #  TypeError, and CropPad then fails on labels=None; this is the construction its Evaluator uses, average_precision_evaluator.py:325)
        random_pad = RandomPadFixedAR(patch_aspect_ratio=img_width / img_height)
        transformations = [convert_to_3_channels, random_pad, resize]
    else:
        raise ValueError("Unexpected argument value: `data_generator_mode` can be either of 'resize' or 'pad', but received '{}'.".format(data_generator_mode))

    generator = data_generator.generate(batch_size=batch_size, shuffle=False, transformations=transformations, label_encoder=None,
                                        returns={'processed_images', 'image_ids', 'inverse_transform'}, keep_images_without_gt=True)
    results = []
    n_images = data_generator.get_dataset_size()
    print("Number of images in the evaluation dataset: {}".format(n_images))
    n_batches = int(ceil(n_images / batch_size))
    try:
        from tqdm import trange
        batches = trange(n_batches, file=sys.stdout)
        batches.set_description('Producing results file')
    except ImportError:                                    # the progress bar is cosmetic
        batches = range(n_batches)
    for _ in batches:
        batch_X, batch_image_ids, batch_inverse_transforms = next(generator)
        y_pred = _predict(model, batch_X)
        if model_mode == 'training':
            from ..ssd_encoder_decoder.ssd_output_decoder import decode_detections
            y_pred = decode_detections(y_pred, confidence_thresh=confidence_thresh, iou_threshold=iou_threshold, top_k=top_k,
                                       input_coords=pred_coords, normalize_coords=normalize_coords, img_height=img_height,
                                       img_width=img_width)
        else:
            if not isinstance(y_pred, np.ndarray):          # a tensor from a torch model's decoding layer
                y_pred = y_pred.detach().float().cpu().numpy()
            y_pred = [y_pred[i][y_pred[i, :, 0] != 0] for i in range(len(y_pred))]      # drop the all-zero dummy rows (:165-169)
        y_pred = apply_inverse_transforms(y_pred, batch_inverse_transforms)              # boxes on the ORIGINAL images
        for k, batch_item in enumerate(y_pred):
            for box in batch_item:
                cat_id = classes_to_cats[box[0]]           # consecutive class id -> original COCO category id
                xmin, ymin = float(round(box[2], 1)), float(round(box[3], 1))
                xmax, ymax = float(round(box[4], 1)), float(round(box[5], 1))
                results.append({'image_id': batch_image_ids[k], 'category_id': cat_id, 'score': float(round(box[1], 3)),
                                'bbox': [xmin, ymin, xmax - xmin, ymax - ymin]})
    with open(out_file, 'w') as f:
        json.dump(results, f)
    print("Prediction results saved in '{}'".format(out_file))
