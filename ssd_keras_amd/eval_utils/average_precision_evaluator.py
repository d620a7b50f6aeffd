"""`Evaluator` -- drop-in for the reference's eval_utils/average_precision_evaluator.py:36-899.

Pascal-VOC-style mean average precision (pre-2010 k-point sampling and post-2010 integration).  The step that
dominates the reference's run time -- `match_predictions`, a Python loop over up to ~184 k predictions per class
(:604-725) -- runs on the GPU (`ssdhip_match_predictions`, csrc/ssdhip_eval.hip: IoU with the reference's mixed
float64 / float32 arithmetic, arg-max, claim resolution by atomic max of the sort key, radix sort, prefix sums).  The
per-class bookkeeping around it (ground truth counts, precision / recall arrays, the 11-point / integrated average) is
the same few NumPy lines as in the reference.

Same constructor, method names, keyword arguments and attributes as the reference.  Deliberate differences:
  * predictions of equal confidence keep their input order (`np.argsort(-conf, kind='mergesort')`); the reference's default
    'quicksort' leaves that order unspecified, so `sorting_algorithm` is accepted and ignored;
  * `verbose=False` matches ALL predictions (the reference then iterates `range(len(predictions.shape))`, i.e. over the
    first prediction of each class only, :650);
  * a class without predictions gets empty cumulative arrays and average precision 0.0 (the reference leaves its
    cumulative lists one entry short, :616-620, and `compute_precision_recall` raises IndexError or mis-assigns classes);
  * `data_generator` is any object with `labels`, `image_ids`, `eval_neutral` (and, for `predict_on_dataset`, a
    `generate(...)` / `get_dataset_size()` pair with the reference generator's contract) -- the reference's own image
    pipeline is outside this package.
"""
from __future__ import annotations

import sys
from math import ceil

import numpy as np

from .. import _native as nat


class Evaluator:
    '''Computes the mean average precision of an SSD model on a dataset (reference class :36-93).'''

    def __init__(self, model, n_classes, data_generator, model_mode='inference',
                 pred_format={'class_id': 0, 'conf': 1, 'xmin': 2, 'ymin': 3, 'xmax': 4, 'ymax': 5},
                 gt_format={'class_id': 0, 'xmin': 1, 'ymin': 2, 'xmax': 3, 'ymax': 4}):
        self.model = model
        self.data_generator = data_generator
        self.n_classes = n_classes
        self.model_mode = model_mode
        self.pred_format = pred_format
        self.gt_format = gt_format
        # per-class lists of length n_classes + 1 (entry 0 is a dummy for the background class), as in the reference
        self.prediction_results = None
        self.num_gt_per_class = None
        self.true_positives = None
        self.false_positives = None
        self.cumulative_true_positives = None
        self.cumulative_false_positives = None
        self.cumulative_precisions = None
        self.cumulative_recalls = None
        self.average_precisions = None
        self.mean_average_precision = None

    def __call__(self, img_height, img_width, batch_size, data_generator_mode='resize', round_confidences=False,
                 matching_iou_threshold=0.5, border_pixels='include', sorting_algorithm='quicksort',
                 average_precision_mode='sample', num_recall_points=11, ignore_neutral_boxes=True, return_precisions=False,
                 return_recalls=False, return_average_precisions=False, verbose=True, decoding_confidence_thresh=0.01,
                 decoding_iou_threshold=0.45, decoding_top_k=200, decoding_pred_coords='centroids',
                 decoding_normalize_coords=True):
        '''Reference :94-256: the whole evaluation in one call.  Returns the mean average precision and, optionally, the
        average precisions, precisions and recalls (in that order, as the reference does).'''
        # the five stages, each leaving its result on the instance exactly as the stand-alone methods do
        self.predict_on_dataset(img_height, img_width, batch_size, data_generator_mode, decoding_confidence_thresh,
                                decoding_iou_threshold, decoding_top_k, decoding_pred_coords, decoding_normalize_coords,
                                border_pixels, round_confidences, verbose)
        self.get_num_gt_per_class(ignore_neutral_boxes, verbose=False)
        self.match_predictions(ignore_neutral_boxes, matching_iou_threshold, border_pixels, sorting_algorithm, verbose)
        self.compute_precision_recall(verbose)
        self.compute_average_precisions(average_precision_mode, num_recall_points, verbose)
        result = [self.compute_mean_average_precision()]
        extras = ((return_average_precisions, self.average_precisions), (return_precisions, self.cumulative_precisions),
                  (return_recalls, self.cumulative_recalls))
        result += [value for wanted, value in extras if wanted]
        return result if len(result) > 1 else result[0]

    def predict_on_dataset(self, img_height, img_width, batch_size, data_generator_mode='resize',
                           decoding_confidence_thresh=0.01, decoding_iou_threshold=0.45, decoding_top_k=200,
                           decoding_pred_coords='centroids', decoding_normalize_coords=True, decoding_border_pixels='include',
                           round_confidences=False, verbose=True, ret=False):
        '''Reference :258-424: run the model over `data_generator` and collect, per class, the tuples
        `(image_id, confidence, xmin, ymin, xmax, ymax)`.  The generator must honour the reference's `generate(batch_size,
        shuffle=False, transformations=..., label_encoder=None, returns={...}, keep_images_without_gt=True,
        degenerate_box_handling='remove')` contract; `transformations` is passed as the mode string ('resize' / 'pad') because
        the reference's image transformation classes are not part of this package.'''
        if data_generator_mode not in ('resize', 'pad'):
            raise ValueError("`data_generator_mode` can be either of 'resize' or 'pad', but received '{}'.".format(data_generator_mode))
        import torch
        from ..ssd_encoder_decoder.ssd_output_decoder import decode_detections
        pf = self.pred_format
        generator = self.data_generator.generate(batch_size=batch_size, shuffle=False, transformations=data_generator_mode,
                                                 label_encoder=None,
                                                 returns={'processed_images', 'image_ids', 'evaluation-neutral', 'inverse_transform',
                                                          'original_labels'},
                                                 keep_images_without_gt=True, degenerate_box_handling='remove')
        if self.data_generator.image_ids is None:
            self.data_generator.image_ids = list(range(self.data_generator.get_dataset_size()))
        results = [list() for _ in range(self.n_classes + 1)]
        n_images = self.data_generator.get_dataset_size()
        n_batches = int(ceil(n_images / batch_size))
        if verbose:
            print("Number of images in the evaluation dataset: {}".format(n_images))
        for _ in range(n_batches):
            batch_X, batch_image_ids, _neutral, batch_inverse_transforms, _orig = next(generator)
            predict = getattr(self.model, 'predict', self.model)
            with torch.no_grad():
                y_pred = predict(batch_X if torch.is_tensor(batch_X) else torch.as_tensor(np.asarray(batch_X), dtype=torch.float32).cuda())
            if self.model_mode == 'training':
                y_pred = decode_detections(y_pred, confidence_thresh=decoding_confidence_thresh, iou_threshold=decoding_iou_threshold,
                                           top_k=decoding_top_k, input_coords=decoding_pred_coords,
                                           normalize_coords=decoding_normalize_coords, img_height=img_height, img_width=img_width,
                                           border_pixels=decoding_border_pixels)
            else:
                y_pred = y_pred.float().cpu().numpy()
                y_pred = [y_pred[i][y_pred[i, :, 0] != 0] for i in range(len(y_pred))]       # drop the zero padding (:395-398)
            if batch_inverse_transforms is not None:                                          # apply_inverse_transforms (:401)
                y_pred = [self._invert(np.copy(y_pred[i]), batch_inverse_transforms[i]) for i in range(len(y_pred))]
            for k, batch_item in enumerate(y_pred):
                image_id = batch_image_ids[k]
                for box in batch_item:
                    class_id = int(box[pf['class_id']])
                    confidence = round(box[pf['conf']], round_confidences) if round_confidences else box[pf['conf']]
                    results[class_id].append((image_id, confidence, round(box[pf['xmin']], 1), round(box[pf['ymin']], 1),
                                              round(box[pf['xmax']], 1), round(box[pf['ymax']], 1)))
        self.prediction_results = results
        if ret:
            return results

    @staticmethod
    def _invert(boxes, inverters):
        for inverter in (inverters or []):
            if inverter is not None and boxes.size:
                boxes = inverter(boxes)
        return boxes

    def write_predictions_to_txt(self, classes=None, out_file_prefix='comp3_det_test_', verbose=True):
        '''Reference :426-475: one Pascal VOC results file per class, a line `image_id confidence xmin ymin xmax ymax` per
        prediction (image id zero-padded to six digits, confidence rounded to four decimals).'''
        if self.prediction_results is None:
            raise ValueError("There are no prediction results. You must run `predict_on_dataset()` before calling this method.")
        for class_id, rows in enumerate(self.prediction_results):
            if class_id == 0:
                continue                                              # entry 0 is the background dummy
            if verbose:
                print("Writing results file for class {}/{}.".format(class_id, self.n_classes))
            name = classes[class_id] if classes is not None else '{:04d}'.format(class_id)
            lines = ['{:06d} {} {}\n'.format(int(row[0]), round(row[1], 4), ' '.join(str(v) for v in row[2:])) for row in rows]
            with open(out_file_prefix + name + '.txt', 'w') as handle:
                handle.writelines(lines)
        if verbose:
            print("All results files saved.")

    def get_num_gt_per_class(self, ignore_neutral_boxes=True, verbose=True, ret=False):
        '''Reference :477-536: number of (non-neutral) ground truth boxes per class over the dataset.'''
        if self.data_generator.labels is None:
            raise ValueError("Computing the number of ground truth boxes per class not possible, no ground truth given.")
        num_gt_per_class = np.zeros(shape=(self.n_classes + 1), dtype=np.int64)
        ci = self.gt_format['class_id']
        neutral = self.data_generator.eval_neutral
        for i, boxes in enumerate(self.data_generator.labels):
            boxes = np.asarray(boxes)
            if boxes.size == 0:
                continue
            cls = boxes[:, ci].astype(np.int64)
            if ignore_neutral_boxes and neutral is not None:
                cls = cls[~np.asarray(neutral[i], dtype=bool)]
            np.add.at(num_gt_per_class, cls, 1)
        self.num_gt_per_class = num_gt_per_class
        if ret:
            return num_gt_per_class

    def match_predictions(self, ignore_neutral_boxes=True, matching_iou_threshold=0.5, border_pixels='include',
                          sorting_algorithm='quicksort', verbose=True, ret=False):
        '''Reference :538-736: match every prediction to the ground truth.  Per class the result arrays are ordered by
        descending confidence: `true_positives[c][i]` / `false_positives[c][i]` flag the i-th most confident prediction,
        `cumulative_*` are their running sums.'''
        if self.data_generator.labels is None:
            raise ValueError("Matching predictions to ground truth boxes not possible, no ground truth given.")
        if self.prediction_results is None:
            raise ValueError("There are no prediction results. You must run `predict_on_dataset()` before calling this method.")
        if border_pixels not in nat.BORDER:
            raise ValueError("`border_pixels` must be one of 'half', 'include' and 'exclude'")
        import torch
        n_classes = self.n_classes
        packed = self._packed_predictions()                     # every class's predictions on the device, packed ONCE per results object
        gt = self._packed_ground_truth(ignore_neutral_boxes)    # ... and the ground truth as CSR over (class, image), once per labels object
        P = int(packed["pred"].shape[0])
        starts = packed["starts_host"]
        if verbose:
            for class_id in range(1, n_classes + 1):
                if starts[class_id] == starts[class_id - 1]:
                    print("No predictions for class {}/{}".format(class_id, n_classes))
            print("Matching predictions to ground truth, classes 1-{} in one launch sequence.".format(n_classes))
            sys.stdout.flush()
        if P:
            # ONE call for all classes (csrc/ssdhip_eval.hip, ssdhip_match_predictions_multi) and ONE download of the five result rows
            res = nat.match_predictions_all(packed["pred"], packed["segment"], packed["slot"], packed["starts"], gt["boxes"], gt["offsets"],
                                            gt["neutral"], matching_iou_threshold, border_pixels).cpu().numpy().astype(np.int64)
        else:
            res = np.zeros((5, 0), dtype=np.int64)
        true_positives, false_positives = [[]], [[]]
        cumulative_true_positives, cumulative_false_positives = [[]], [[]]
        for class_id in range(1, n_classes + 1):
            lo, hi = int(starts[class_id - 1]), int(starts[class_id])
            true_positives.append(res[1, lo:hi].copy())
            false_positives.append(res[2, lo:hi].copy())
            cumulative_true_positives.append(res[3, lo:hi].copy())
            cumulative_false_positives.append(res[4, lo:hi].copy())
        self.true_positives = true_positives
        self.false_positives = false_positives
        self.cumulative_true_positives = cumulative_true_positives
        self.cumulative_false_positives = cumulative_false_positives
        if ret:
            return true_positives, false_positives, cumulative_true_positives, cumulative_false_positives

    # ---- packing: the reference keeps predictions as per-class Python lists of tuples and labels as a list of arrays; walking them
    #      is host work that dwarfs the matching itself (94 k predictions: ~70 ms of tuple handling for ~1 ms of kernels).  Both are
    #      packed once per OBJECT and kept on the device: a second evaluation of the same results (another IoU threshold, another border
    #      mode) only launches.  The cache keys on the identity of `prediction_results` / `labels` and on the per-class lengths: whoever
    #      edits those lists IN PLACE without changing a length calls `forget_packed_inputs()`. --------------------------------------------
    def forget_packed_inputs(self):
        """Drop the device copies of the predictions and the ground truth (rebuilt by the next `match_predictions`)."""
        for name in ("_packed_pred_memo", "_packed_gt_memo", "_image_index_memo"):
            self.__dict__.pop(name, None)

    def _image_index(self):
        ids = self.data_generator.image_ids
        memo = self.__dict__.get("_image_index_memo")
        if memo is None or memo[0] is not ids or memo[1] != len(ids):
            memo = (ids, len(ids), {str(v): i for i, v in enumerate(ids)})
            self.__dict__["_image_index_memo"] = memo
        return memo[2]

    def _packed_predictions(self):
        import torch
        results = self.prediction_results
        sig = (id(results), tuple(len(r) for r in results), id(self.data_generator.image_ids))
        memo = self.__dict__.get("_packed_pred_memo")
        if memo is not None and memo["sig"] == sig and memo["results"] is results:
            return memo
        n_classes = self.n_classes
        n_images = len(self.data_generator.image_ids)
        index_of = self._image_index()
        counts = np.array([len(results[c]) for c in range(1, n_classes + 1)], dtype=np.int64)
        starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        P = int(starts[-1])
        pred = np.empty((P, 5), dtype=np.float32)                 # 'f4' like the reference's structured array (:629-634)
        image = np.empty((P,), dtype=np.int64)
        for c in range(1, n_classes + 1):
            rows = results[c]
            if not rows:
                continue
            lo, hi = int(starts[c - 1]), int(starts[c])
            columns = list(zip(*rows))                            # (image ids, conf, xmin, ymin, xmax, ymax)
            for j in range(5):
                pred[lo:hi, j] = columns[j + 1]
            image[lo:hi] = [index_of[str(v)] for v in columns[0]]
        slot = np.repeat(np.arange(n_classes, dtype=np.int64), counts)
        dev = torch.device("cuda", torch.cuda.current_device())
        memo = {"sig": sig, "results": results, "starts_host": starts,
                "pred": nat.to_device(pred, device=dev, dtype=torch.float32),
                "segment": nat.to_device((slot * n_images + image).astype(np.int32), device=dev, dtype=torch.int32),
                "slot": nat.to_device(slot.astype(np.int32), device=dev, dtype=torch.int32),
                "starts": nat.to_device(starts.astype(np.int32), device=dev, dtype=torch.int32)}
        self.__dict__["_packed_pred_memo"] = memo
        return memo

    def _packed_ground_truth(self, ignore_neutral_boxes):
        import torch
        labels = self.data_generator.labels
        neutral = self.data_generator.eval_neutral if (ignore_neutral_boxes and self.data_generator.eval_neutral is not None) else None
        gf = self.gt_format
        sig = (id(labels), len(labels), id(neutral), self.n_classes, tuple(sorted(gf.items())))
        memo = self.__dict__.get("_packed_gt_memo")
        if memo is not None and memo["sig"] == sig and memo["labels"] is labels:
            return memo
        cols = [gf['xmin'], gf['ymin'], gf['xmax'], gf['ymax']]
        n_images, n_classes = len(labels), self.n_classes
        gt_cat, gt_img, neutral_cat = self._concat_ground_truth(labels, neutral)
        if gt_cat.shape[0]:
            cls = gt_cat[:, gf['class_id']].astype(np.int64)
            keep = (cls >= 1) & (cls <= n_classes)                # rows of other classes are never looked at (the reference masks per class)
            seg = (cls - 1) * n_images + gt_img
            order = np.argsort(np.where(keep, seg, n_classes * n_images), kind='stable')[:int(keep.sum())]   # image order inside a segment
            boxes = gt_cat[order][:, cols].astype(np.float64)
            per_segment = np.bincount(seg[order], minlength=n_classes * n_images)
            flags = neutral_cat[order].astype(np.uint8) if neutral_cat is not None else None
        else:
            boxes, per_segment, flags = np.zeros((0, 4)), np.zeros(n_classes * n_images, dtype=np.int64), None
        offsets = np.concatenate([[0], np.cumsum(per_segment)]).astype(np.int32)
        dev = torch.device("cuda", torch.cuda.current_device())
        memo = {"sig": sig, "labels": labels,
                "boxes": nat.to_device(np.ascontiguousarray(boxes), device=dev, dtype=torch.float64),
                "offsets": nat.to_device(offsets, device=dev, dtype=torch.int32),
                "neutral": nat.to_device(flags, device=dev, dtype=torch.uint8) if flags is not None else None}
        self.__dict__["_packed_gt_memo"] = memo
        return memo

    @staticmethod
    def _concat_ground_truth(labels, neutral):
        """All label arrays stacked once: (G_all, n_cols) rows in image order, the image index of each row, neutral flags or None."""
        arrs = [np.asarray(lab) for lab in labels]
        counts = np.array([a.shape[0] if a.ndim == 2 else 0 for a in arrs], dtype=np.int64)
        keep = [a for a, c in zip(arrs, counts) if c]
        cat = np.concatenate(keep, axis=0) if keep else np.zeros((0, 5))
        img = np.repeat(np.arange(len(arrs), dtype=np.int64), counts)
        ncat = None
        if neutral is not None:
            flags = [np.asarray(neutral[i], dtype=bool).reshape(-1) for i, c in enumerate(counts) if c]
            ncat = np.concatenate(flags) if flags else np.zeros((0,), dtype=bool)
        return cat, img, ncat

    @staticmethod
    def _class_ground_truth(gt_cat, gt_img, neutral_cat, n_images, class_id, class_col, cols):
        """One class's ground truth as CSR over the images: boxes (G,4) float64 (integer labels convert exactly), offsets
        (n_images + 1,) int32, neutral flags (G,) uint8 or None.  Rows stay in image order, as the reference's per-image masks."""
        mask = gt_cat[:, class_col] == class_id if gt_cat.shape[0] else np.zeros((0,), dtype=bool)
        boxes = gt_cat[mask][:, cols].astype(np.float64) if gt_cat.shape[0] else np.zeros((0, 4))
        per_image = np.bincount(gt_img[mask], minlength=n_images) if gt_cat.shape[0] else np.zeros(n_images, dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(per_image)]).astype(np.int32)
        flags = neutral_cat[mask].astype(np.uint8) if neutral_cat is not None else None
        return boxes, offsets, flags

    def compute_precision_recall(self, verbose=True, ret=False):
        '''Reference :738-781: per class, precision = TP / (TP + FP) (0 where nothing has been predicted yet) and
        recall = TP / #ground truth, over the confidence-sorted running counts.'''
        if (self.cumulative_true_positives is None) or (self.cumulative_false_positives is None):
            raise ValueError("True and false positives not available. You must run `match_predictions()` before you call this method.")
        if self.num_gt_per_class is None:
            raise ValueError("Number of ground truth boxes per class not available. You must run `get_num_gt_per_class()` before "
                             "you call this method.")
        precisions, recalls = [[]], [[]]
        for class_id in range(1, self.n_classes + 1):
            if verbose:
                print("Computing precisions and recalls, class {}/{}".format(class_id, self.n_classes))
            hits = self.cumulative_true_positives[class_id]
            seen = hits + self.cumulative_false_positives[class_id]
            with np.errstate(divide='ignore', invalid='ignore'):
                ratio = hits / seen
                recalls.append(hits / self.num_gt_per_class[class_id])
            ratio[seen <= 0] = 0
            precisions.append(ratio)
        self.cumulative_precisions, self.cumulative_recalls = precisions, recalls
        if ret:
            return precisions, recalls

    @staticmethod
    def _average_precision(precision, recall, mode, num_recall_points):
        """Average precision of one class from its precision / recall arrays (recall is non-decreasing).  The best precision
        at recall >= r is a suffix maximum of `precision`, so both Pascal VOC variants are a reversed running maximum plus a
        sorted search instead of a loop over thresholds / recall levels:
          'sample'    (pre-2010, reference :829-842): mean over num_recall_points thresholds t of that maximum at the first
                      position whose recall reaches t (0 when none does), accumulated in threshold order like the reference;
          'integrate' (post-2010, reference :844-881): sum over the distinct recall levels but the last of
                      (next level - level) x (best precision from the level's first position up to the last level's first)."""
        n = len(precision)
        if n == 0:
            return 0.0
        if mode == 'sample' and np.isnan(recall).any():
            # a class without ground truth boxes: recall is 0/0 until the first hit, so it is not sorted -- the definition itself
            total = 0.0
            for t in np.linspace(0, 1, num_recall_points):
                reached = precision[recall >= t]
                total += reached.max() if reached.size else 0.0
            return total / num_recall_points
        if mode == 'sample':
            best_from = np.maximum.accumulate(precision[::-1])[::-1]
            first = np.searchsorted(recall, np.linspace(0, 1, num_recall_points), side='left')
            total = 0.0
            for i in first:
                total += best_from[i] if i < n else 0.0
            return total / num_recall_points
        levels, starts = np.unique(recall, return_index=True)
        best, width = np.zeros_like(levels), np.zeros_like(levels)
        if len(levels) > 1:
            head = precision[:starts[-1]]
            best[:-1] = np.maximum(np.maximum.accumulate(head[::-1])[::-1][starts[:-1]], 0.0)
            width[:-1] = levels[1:] - levels[:-1]
        return np.sum(best * width)

    def compute_average_precisions(self, mode='sample', num_recall_points=11, verbose=True, ret=False):
        '''Reference :783-884: 'sample' = Pascal VOC pre-2010 k-point sampling, 'integrate' = post-2010 integration.'''
        if (self.cumulative_precisions is None) or (self.cumulative_recalls is None):
            raise ValueError("Precisions and recalls not available. You must run `compute_precision_recall()` before you call this method.")
        if mode not in {'sample', 'integrate'}:
            raise ValueError("`mode` can be either 'sample' or 'integrate', but received '{}'".format(mode))
        per_class = [0.0]
        for class_id in range(1, self.n_classes + 1):
            if verbose:
                print("Computing average precision, class {}/{}".format(class_id, self.n_classes))
            per_class.append(self._average_precision(np.asarray(self.cumulative_precisions[class_id], dtype=np.float64),
                                                     np.asarray(self.cumulative_recalls[class_id], dtype=np.float64), mode,
                                                     num_recall_points))
        self.average_precisions = per_class
        if ret:
            return per_class

    def compute_mean_average_precision(self, ret=True):
        '''Reference :886-899: the mean over the positive classes (entry 0 is the background dummy).'''
        if self.average_precisions is None:
            raise ValueError("Average precisions not available. You must run `compute_average_precisions()` before you call this method.")
        self.mean_average_precision = np.average(self.average_precisions[1:])
        if ret:
            return self.mean_average_precision
