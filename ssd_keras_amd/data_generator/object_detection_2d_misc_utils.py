"""Drop-in for the reference's data_generator/object_detection_2d_misc_utils.py: `apply_inverse_transforms` :22-73 -- maps
decoded predictions made on transformed images back to the original images with the inverter functions the image
transformations return (e.g. `CropPad`'s).  Host-side bookkeeping between `decode_detections` and its callers
(`Evaluator.predict_on_dataset`, `coco_utils.predict_all_to_json`)."""
from __future__ import annotations

import numpy as np


def apply_inverse_transforms(y_pred_decoded, inverse_transforms):
    '''`y_pred_decoded`: a list of `(num_predictions, 6)` arrays or one `(batch_size, num_predictions, 6)` array;
    `inverse_transforms[i]`: the inverter functions of batch item i, applied in order (`None` entries are skipped, batch items
    without predictions are left alone).  Returns a copy with the same structure.'''
    if isinstance(y_pred_decoded, list):
        inverted = [np.copy(item) for item in y_pred_decoded]
    elif isinstance(y_pred_decoded, np.ndarray):
        inverted = np.copy(y_pred_decoded)
    else:
        raise ValueError("`y_pred_decoded` must be either a list or a Numpy array.")
    for i in range(len(y_pred_decoded)):
        if inverted[i].size > 0:
            for inverter in inverse_transforms[i]:
                if inverter is not None:
                    inverted[i] = inverter(inverted[i])
    return inverted
