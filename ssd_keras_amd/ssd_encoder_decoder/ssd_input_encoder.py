"""`SSDInputEncoder` -- drop-in for the reference's ssd_encoder_decoder/ssd_input_encoder.py:25-611.

Same constructor arguments, validation errors, public attributes (`boxes_list`, `n_classes`, `*_diag`) and
`__call__` contract as the reference.  Anchors are generated once on the host in float64 (configuration
work); the per-batch work -- pairwise IoU, bipartite + multi matching, neutral marking, one-hot / offset
encoding and the template itself (`generate_encoding_template`, 2/3 of the reference's encoder time) -- runs
in libssdhip.so (`ssdhip_encode`, three kernels) with the anchors resident in HBM.

`__call__` returns a NumPy float64 array like the reference (that copy crosses PCIe);
`encode_to_device` returns the float32 CUDA tensor a training step consumes, with no host round trip.
"""
from __future__ import annotations

import ctypes

import numpy as np

from .. import _native as nat
from ..anchor_math import layer_anchor_boxes


class DegenerateBoxError(Exception):
    '''An exception class to be raised if degenerate boxes are being detected (reference :613).'''
    pass


class SSDInputEncoder:
    def __init__(self, img_height, img_width, n_classes, predictor_sizes, min_scale=0.1, max_scale=0.9, scales=None,
                 aspect_ratios_global=[0.5, 1.0, 2.0], aspect_ratios_per_layer=None, two_boxes_for_ar1=True, steps=None,
                 offsets=None, clip_boxes=False, variances=[0.1, 0.1, 0.2, 0.2], matching_type='multi',
                 pos_iou_threshold=0.5, neg_iou_limit=0.3, border_pixels='half', coords='centroids',
                 normalize_coords=True, background_id=0):
        predictor_sizes = np.array(predictor_sizes)
        if predictor_sizes.ndim == 1:
            predictor_sizes = np.expand_dims(predictor_sizes, axis=0)
        L = predictor_sizes.shape[0]
        # ---- the reference's argument checks (:142-180), same exception types ----
        if (min_scale is None or max_scale is None) and scales is None:
            raise ValueError("Either `min_scale` and `max_scale` or `scales` need to be specified.")
        if scales is not None and len(scales):
            if len(scales) != L + 1:
                raise ValueError("It must be either scales is None or len(scales) == len(predictor_sizes)+1, but len(scales) == {} "
                                 "and len(predictor_sizes)+1 == {}".format(len(scales), L + 1))
            scales = np.array(scales)
            if np.any(scales <= 0):
                raise ValueError("All values in `scales` must be greater than 0, but the passed list of scales is {}".format(scales))
        else:
            scales = None
            if not 0 < min_scale <= max_scale:
                raise ValueError("It must be 0 < min_scale <= max_scale, but it is min_scale = {} and max_scale = {}".format(min_scale, max_scale))
        if aspect_ratios_per_layer is not None:
            if len(aspect_ratios_per_layer) != L:
                raise ValueError("It must be either aspect_ratios_per_layer is None or len(aspect_ratios_per_layer) == "
                                 "len(predictor_sizes), but len(aspect_ratios_per_layer) == {} and len(predictor_sizes) == {}".format(
                                     len(aspect_ratios_per_layer), L))
            for ar in aspect_ratios_per_layer:
                if np.any(np.array(ar) <= 0):
                    raise ValueError("All aspect ratios must be greater than zero.")
        else:
            if aspect_ratios_global is None:
                raise ValueError("At least one of `aspect_ratios_global` and `aspect_ratios_per_layer` must not be `None`.")
            if np.any(np.array(aspect_ratios_global) <= 0):
                raise ValueError("All aspect ratios must be greater than zero.")
        if len(variances) != 4:
            raise ValueError("4 variance values must be pased, but {} values were received.".format(len(variances)))
        variances = np.array(variances)
        if np.any(variances <= 0):
            raise ValueError("All variances must be >0, but the variances given are {}".format(variances))
        if coords not in ('minmax', 'centroids', 'corners'):
            raise ValueError("Unexpected value for `coords`. Supported values are 'minmax', 'corners' and 'centroids'.")
        if (steps is not None) and (len(steps) != L):
            raise ValueError("You must provide at least one step value per predictor layer.")
        if (offsets is not None) and (len(offsets) != L):
            raise ValueError("You must provide at least one offset value per predictor layer.")
        if matching_type not in ('multi', 'bipartite'):
            raise ValueError("`matching_type` must be 'multi' or 'bipartite'.")
        if border_pixels not in nat.BORDER:
            raise ValueError("`border_pixels` must be 'half', 'include' or 'exclude'.")

        self.img_height, self.img_width = img_height, img_width
        self.n_classes = n_classes + 1                     # + background, as the reference (:186)
        self.predictor_sizes = predictor_sizes
        self.min_scale, self.max_scale = min_scale, max_scale
        self.scales = np.linspace(min_scale, max_scale, L + 1) if scales is None else scales
        self.aspect_ratios = [aspect_ratios_global] * L if aspect_ratios_per_layer is None else aspect_ratios_per_layer
        self.two_boxes_for_ar1 = two_boxes_for_ar1
        self.steps = steps if steps is not None else [None] * L
        self.offsets = offsets if offsets is not None else [None] * L
        self.clip_boxes = clip_boxes
        self.variances = variances
        self.matching_type = matching_type
        self.pos_iou_threshold = pos_iou_threshold
        self.neg_iou_limit = neg_iou_limit
        self.border_pixels = border_pixels
        self.coords = coords
        self.normalize_coords = normalize_coords
        self.background_id = background_id
        if aspect_ratios_per_layer is not None:
            self.n_boxes = [len(ar) + 1 if (1 in ar) and two_boxes_for_ar1 else len(ar) for ar in aspect_ratios_per_layer]
        else:
            self.n_boxes = len(aspect_ratios_global) + 1 if (1 in aspect_ratios_global) and two_boxes_for_ar1 else len(aspect_ratios_global)

        self.boxes_list, self.wh_list_diag, self.steps_diag, self.offsets_diag, self.centers_diag = [], [], [], [], []
        for i in range(L):
            boxes, center, wh, step, offset = self.generate_anchor_boxes_for_layer(
                feature_map_size=self.predictor_sizes[i], aspect_ratios=self.aspect_ratios[i], this_scale=self.scales[i],
                next_scale=self.scales[i + 1], this_steps=self.steps[i], this_offsets=self.offsets[i], diagnostics=True)
            self.boxes_list.append(boxes)
            self.wh_list_diag.append(wh)
            self.steps_diag.append(step)
            self.offsets_diag.append(offset)
            self.centers_diag.append(center)
        self._anchors_host = np.ascontiguousarray(np.concatenate([b.reshape(-1, 4) for b in self.boxes_list], axis=0))
        self._dev = {}

    # ------------------------------------------------------------------------------------------
    def generate_anchor_boxes_for_layer(self, feature_map_size, aspect_ratios, this_scale, next_scale, this_steps=None,
                                        this_offsets=None, diagnostics=False):
        '''Reference :420-548; host float64 (see anchor_math.layer_anchor_boxes).'''
        return layer_anchor_boxes(self.img_height, self.img_width, feature_map_size, aspect_ratios, this_scale, next_scale,
                                  self.two_boxes_for_ar1, this_steps, this_offsets, self.clip_boxes, self.coords,
                                  self.normalize_coords, diagnostics=diagnostics)

    @property
    def n_anchors(self):
        return self._anchors_host.shape[0]

    def __getstate__(self):
        """Pickling / deep copies (DataLoader workers, a copied generator configuration) carry the host-side configuration only: the
        resident device constants and the pinned upload ring (pinned tensors, HIP events) are per-process and rebuilt on first use."""
        state = dict(self.__dict__)
        state.pop('_pinned_ring', None)
        state['_dev'] = {}
        return state

    def _device_constants(self, device):
        import torch
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self._anchors_host).to(device),
                              torch.from_numpy(np.asarray(self.variances, dtype=np.float64)).to(device))
        return self._dev[key]

    def _pack_ground_truth(self, ground_truth_labels):
        '''Host-side checks of :327-336 (labels originate on the host) + CSR packing.  One concatenation and vectorised
        checks over the whole batch: the per-image Python work is an `asarray` and a `reshape`.'''
        rows, counts = [], np.zeros(len(ground_truth_labels) + 1, dtype=np.int64)
        for i, lab in enumerate(ground_truth_labels):
            lab = np.asarray(lab)
            if lab.size:
                lab = lab.reshape(-1, 5)
                rows.append(lab)
                counts[i + 1] = lab.shape[0]
        offsets = np.cumsum(counts)
        if not rows:
            return np.zeros((0, 5)), offsets.astype(np.int32), 0
        gt = np.ascontiguousarray(np.concatenate(rows, axis=0), dtype=np.float64)
        bad = (gt[:, 3] - gt[:, 1] <= 0) | (gt[:, 4] - gt[:, 2] <= 0)
        if bad.any():
            i = int(np.searchsorted(offsets, int(np.argmax(bad)), side='right')) - 1
            lab = gt[offsets[i]:offsets[i + 1]]
            raise DegenerateBoxError("SSDInputEncoder detected degenerate ground truth bounding boxes for batch item {} with "
                                     "bounding boxes {}, ".format(i, lab) + "i.e. bounding boxes where xmax <= xmin and/or "
                                     "ymax <= ymin. Degenerate ground truth bounding boxes will lead to NaN errors during "
                                     "the training.")
        cls = gt[:, 0].astype(np.int64)
        if cls.min() < 0 or cls.max() >= self.n_classes:
            raise IndexError("class id out of range for {} classes (incl. background)".format(self.n_classes))
        return gt, offsets.astype(np.int32), int(counts.max())

    def encode_to_device(self, ground_truth_labels, device=None, want_f32=True, want_f64=False, want_matches=False):
        '''Run the encoder kernels; outputs stay in HBM.  Returns (y_f32 | None, y_f64 | None, match_gt | None).'''
        import torch
        lib = nat.load()
        if not hasattr(lib, 'ssdhip_encode'):
            raise nat.SsdHipError("libssdhip.so was built without the encoder")
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        gt, offsets, max_g = self._pack_ground_truth(ground_truth_labels)
        if max_g > 1024:
            raise ValueError("at most 1024 ground truth boxes per image are supported, got {}".format(max_g))
        B = len(ground_truth_labels)
        # one upload: [offsets int32 (B+1), padded to 8 bytes | ground truth rows float64]
        n_off = (offsets.shape[0] + 1) // 2
        packed = np.empty(n_off + max(int(gt.shape[0]), 1) * 5, dtype=np.float64)
        packed[:n_off].view(np.int32)[:offsets.shape[0]] = offsets
        packed[n_off:n_off + gt.size] = gt.ravel()
        packed_d = self._upload(packed, device)
        return self.encode_packed(packed_d[n_off:], packed_d[:n_off].view(torch.int32), int(gt.shape[0]), int(max_g), B, want_f32,
                                  want_f64, want_matches)

    def _upload(self, packed, device):
        """The packed labels -> HBM WITHOUT stalling the host.  `torch.from_numpy(a).to(device)` of a pageable array returns only when
        the copy has run, and the copy queues behind everything the stream still holds: in a training loop that is a full device
        synchronisation per step -- the host could not issue the next forward pass while the GPU finished the last backward (round 5:
        ~0.5 ms of an 11 ms step).  A small ring of pinned staging buffers + an asynchronous copy instead; a slot is reused only after
        its own copy has completed (an event per slot: four steps back, long done)."""
        import os
        import torch
        if os.environ.get("SSDHIP_SYNC_UPLOAD", "0") == "1":               # the round-4 form, for A/B timing
            return torch.from_numpy(packed).to(device)
        ring = self.__dict__.setdefault('_pinned_ring', {}).setdefault(str(device), {'slots': [None] * 4, 'next': 0})
        i = ring['next']
        ring['next'] = (i + 1) % len(ring['slots'])
        slot = ring['slots'][i]
        n = int(packed.shape[0])
        if slot is not None:
            slot[1].synchronize()
        if slot is None or slot[0].numel() < n:
            slot = [torch.empty((max(n, 1024),), dtype=torch.float64).pin_memory(), None]
        slot[0].numpy()[:n] = packed
        out = torch.empty((n,), dtype=torch.float64, device=device)
        with torch.cuda.device(device):
            out.copy_(slot[0][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        slot[1] = ev
        ring['slots'][i] = slot
        return out

    def encode_packed(self, gt_d, off_d, n_gt, max_gt_per_image, batch_size, want_f32=True, want_f64=False, want_matches=False):
        '''The encoder kernels on labels that are ALREADY on the GPU in CSR form (e.g. produced by a device-side input pipeline):
        `gt_d` float64 (n_gt, 5) rows `[class, xmin, ymin, xmax, ymax]` of all images concatenated, `off_d` int32 (batch + 1,)
        row offsets.  No host work, no PCIe traffic, nothing but `ssdhip_encode`'s two launches on the current stream; the caller
        vouches for the checks `encode_to_device` makes on the host (no degenerate boxes, class ids in range, <= 1024 boxes per
        image).  `max_gt_per_image` must be an UPPER bound of the per-image box counts in `off_d`: the kernels size their tables
        with it and ignore an image's boxes beyond it (they never index past the tables; the targets of such an image are then
        wrong, not the memory).  Returns (y_f32 | None, y_f64 | None, match_gt | None).'''
        import torch
        lib = nat.load()
        device = off_d.device
        B, N, C = int(batch_size), self.n_anchors, self.n_classes
        anchors, variances = self._device_constants(device)
        y32 = torch.empty((B, N, C + 12), dtype=torch.float32, device=device) if want_f32 else None
        y64 = torch.empty((B, N, C + 12), dtype=torch.float64, device=device) if want_f64 else None
        mm = torch.empty((B, N), dtype=torch.int32, device=device) if want_matches else None
        need = lib.ssdhip_encode_workspace_bytes(B, N, C, int(n_gt))
        ws = nat.workspaces.get(device, 'encode', need)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(device):
            rc = lib.ssdhip_encode(ptr(anchors), ptr(variances), ptr(gt_d), ptr(off_d), int(n_gt), int(max_gt_per_image), B, N, C,
                                   float(self.img_height), float(self.img_width), 1 if self.matching_type == 'multi' else 0,
                                   float(self.pos_iou_threshold), float(self.neg_iou_limit), nat.COORDS[self.coords],
                                   int(bool(self.normalize_coords)), nat.BORDER[self.border_pixels], int(self.background_id),
                                   ptr(y32), ptr(y64), ptr(mm), ptr(ws), ws.numel(), nat.current_stream_ptr(device))
        nat.check(rc, 'ssdhip_encode')
        return y32, y64, mm

    def __call__(self, ground_truth_labels, diagnostics=False):
        '''Reference :277-418.  Returns `y_encoded` (batch, #boxes, #classes + 12) float64 [and, with `diagnostics`,
        the copy whose offsets are zeroed (:412-416)].'''
        _, y64, _ = self.encode_to_device(ground_truth_labels, want_f32=False, want_f64=True)
        y_encoded = y64.cpu().numpy()
        if diagnostics:
            y_matched_anchors = np.copy(y_encoded)
            y_matched_anchors[:, :, -12:-8] = 0
            return y_encoded, y_matched_anchors
        return y_encoded

    def generate_encoding_template(self, batch_size, diagnostics=False):
        '''Reference :550-611: [zeros C | anchors | anchors | variances] float64 (host; constant per configuration).'''
        a = self._anchors_host
        t = np.zeros((batch_size, a.shape[0], self.n_classes + 12))
        t[:, :, self.n_classes:self.n_classes + 4] = a
        t[:, :, self.n_classes + 4:self.n_classes + 8] = a
        t[:, :, self.n_classes + 8:] = self.variances
        if diagnostics:
            return t, self.centers_diag, self.wh_list_diag, self.steps_diag, self.offsets_diag
        return t
