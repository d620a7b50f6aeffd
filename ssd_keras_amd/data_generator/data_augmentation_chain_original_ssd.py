"""Drop-in for the reference's data_generator/data_augmentation_chain_original_ssd.py: `SSDRandomCrop`
:29-101 and `SSDExpand` :103-162 -- the random crops (Caffe SSD `batch_sampler`) and the random expansion (`train_transform_param`)
of the original SSD training pipeline, both configurations of the patch sampling ops with the IoU validation of the candidate crops on
the GPU (object_detection_2d_patch_sampling_ops.py) -- `SSDPhotometricDistortions` :146-208 and the whole `SSDDataAugmentation` chain
:210-280 (photometric distortions -> expansion -> random crop -> random flip -> resize with a random interpolation mode).

The photometric distortions of one image are ONE kernel launch: the chain's random draws are made first, in the reference's order, and
turned into a per-image program of pointwise steps (csrc/ssdhip_image.hip) -- with its uint8 <-> float32 round trips and their
roundings, exactly as the reference's list of transforms would apply them one NumPy pass at a time.  `distort_batch` does the same for
a CUDA batch: one draw sequence per image, one launch for all of them.
"""
from __future__ import annotations

import inspect
import os

import numpy as np

from . import _image_ops as iop
from .object_detection_2d_geometric_ops import (INTER_AREA, INTER_CUBIC, INTER_LANCZOS4, INTER_LINEAR, INTER_NEAREST, RandomFlip,
                                                ResizeRandomInterp)
from .object_detection_2d_image_boxes_validation_utils import BoundGenerator, BoxFilter, ImageValidator
from .object_detection_2d_patch_sampling_ops import PatchCoordinateGenerator, RandomPatch, RandomPatchInf
from .object_detection_2d_photometric_ops import (ConvertTo3Channels, RandomBrightness, RandomChannelSwap, RandomContrast, RandomHue,
                                                  RandomSaturation)

_DEFAULT_FORMAT = {'class_id': 0, 'xmin': 1, 'ymin': 2, 'xmax': 3, 'ymax': 4}


class SSDRandomCrop:
    '''The random crops of the original SSD: per round one of six lower IoU bounds (none, 0.1 ... 0.9) is drawn, up to 50 candidate
    patches of 0.3-1.0 of the image size and aspect ratio 0.5-2.0 are tried, a patch is valid if at least one ground truth box
    overlaps it with an IoU above the bound; boxes whose centre falls outside the chosen patch are dropped; with probability
    0.143 per round the image is returned as it is (reference :29-101).'''

    def __init__(self, labels_format=_DEFAULT_FORMAT):
        self.labels_format = labels_format
        self.bound_generator = BoundGenerator(sample_space=((None, None), (0.1, None), (0.3, None), (0.5, None), (0.7, None), (0.9, None)),
                                              weights=None)
        self.patch_coord_generator = PatchCoordinateGenerator(must_match='h_w', min_scale=0.3, max_scale=1.0, scale_uniformly=False,
                                                              min_aspect_ratio=0.5, max_aspect_ratio=2.0)
        self.box_filter = BoxFilter(check_overlap=True, check_min_area=False, check_degenerate=False, overlap_criterion='center_point',
                                    labels_format=self.labels_format)
        self.image_validator = ImageValidator(overlap_criterion='iou', n_boxes_min=1, labels_format=self.labels_format,
                                              border_pixels='half')
        self.random_crop = RandomPatchInf(patch_coord_generator=self.patch_coord_generator, box_filter=self.box_filter,
                                          image_validator=self.image_validator, bound_generator=self.bound_generator, n_trials_max=50,
                                          clip_boxes=True, prob=0.857, labels_format=self.labels_format)

    def __call__(self, image, labels=None, return_inverter=False):
        self.random_crop.labels_format = self.labels_format
        return self.random_crop(image, labels, return_inverter)


class SSDExpand:
    '''The random expansion of the original SSD: with probability 0.5 the image is placed at a random position on a canvas of
    1-4 times its size filled with the mean colour (reference :103-162).'''

    def __init__(self, background=(123, 117, 104), labels_format=_DEFAULT_FORMAT):
        self.labels_format = labels_format
        self.patch_coord_generator = PatchCoordinateGenerator(must_match='h_w', min_scale=1.0, max_scale=4.0, scale_uniformly=True)
        self.expand = RandomPatch(patch_coord_generator=self.patch_coord_generator, box_filter=None, image_validator=None, n_trials_max=1,
                                  clip_boxes=False, prob=0.5, background=background, labels_format=self.labels_format)

    def __call__(self, image, labels=None, return_inverter=False):
        self.expand.labels_format = self.labels_format
        return self.expand(image, labels, return_inverter)


class SSDPhotometricDistortions:
    '''The photometric distortions of the original SSD (`train_transform_param`): random brightness (+-32), contrast (0.5-1.5),
    saturation (0.5-1.5) and hue (+-18), each with probability 0.5, contrast before or after the HSV part with probability 0.5 each,
    never a channel swap (reference :146-208).'''

    def __init__(self):
        self.convert_to_3_channels = ConvertTo3Channels()
        self.random_brightness = RandomBrightness(lower=-32, upper=32, prob=0.5)
        self.random_contrast = RandomContrast(lower=0.5, upper=1.5, prob=0.5)
        self.random_saturation = RandomSaturation(lower=0.5, upper=1.5, prob=0.5)
        self.random_hue = RandomHue(max_delta=18, prob=0.5)
        self.random_channel_swap = RandomChannelSwap(prob=0.0)

    def draw(self):
        """The random draws of one call, in the reference's order, as the program of pointwise steps they select."""
        hsv_part = lambda: ([("to_u8", 0), ("rgb2hsv", 0), ("to_f32", 0)] + self.random_saturation.draw() + self.random_hue.draw()
                            + [("to_u8", 0), ("hsv2rgb", 0)])
        if np.random.choice(2):                                            # sequence 1: contrast before the HSV part (:164-175)
            steps = [("to_f32", 0)] + self.random_brightness.draw() + self.random_contrast.draw() + hsv_part()
            steps += self.random_channel_swap.draw()
        else:                                                              # sequence 2: contrast after it (:177-190)
            steps = [("to_f32", 0)] + self.random_brightness.draw() + hsv_part()
            steps += [("to_f32", 0)] + self.random_contrast.draw() + [("to_u8", 0)] + self.random_channel_swap.draw()
        return steps

    def __call__(self, image, labels):
        image, labels = self.convert_to_3_channels(image, labels)
        return iop.run(image, self.draw()), labels

    def distort_batch(self, images):
        """images (B, H, W, 3) CUDA uint8 or float32 -> distorted uint8 batch: one draw sequence per image (the order a loop over the
        reference's chain would make them in), ONE launch."""
        return iop.run_batch(images, [self.draw() for _ in range(int(images.shape[0]))])


class SSDDataAugmentation:
    '''The data augmentation pipeline of the original Caffe SSD (reference :210-280): photometric distortions, expansion, random crop,
    random horizontal flip, resize to the network input with a random interpolation mode (degenerate boxes dropped).'''

    def __init__(self, img_height=300, img_width=300, background=(123, 117, 104), labels_format=_DEFAULT_FORMAT):
        self.labels_format = labels_format
        self.photometric_distortions = SSDPhotometricDistortions()
        self.expand = SSDExpand(background=background, labels_format=self.labels_format)
        self.random_crop = SSDRandomCrop(labels_format=self.labels_format)
        self.random_flip = RandomFlip(dim='horizontal', prob=0.5, labels_format=self.labels_format)
        # resizing can shrink small boxes to zero height / width: those are dropped
        self.box_filter = BoxFilter(check_overlap=False, check_min_area=False, check_degenerate=True, labels_format=self.labels_format)
        self.resize = ResizeRandomInterp(height=img_height, width=img_width,
                                         interpolation_modes=[INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4],
                                         box_filter=self.box_filter, labels_format=self.labels_format)
        self.sequence = [self.photometric_distortions, self.expand, self.random_crop, self.random_flip, self.resize]

    def __call__(self, image, labels, return_inverter=False):
        self.expand.labels_format = self.labels_format
        self.random_crop.labels_format = self.labels_format
        self.random_flip.labels_format = self.labels_format
        self.resize.labels_format = self.labels_format
        inverters = []
        for transform in self.sequence:
            if return_inverter and ('return_inverter' in inspect.signature(transform).parameters):
                image, labels, inverter = transform(image, labels, return_inverter=True)
                inverters.append(inverter)
            else:
                image, labels = transform(image, labels)
        return (image, labels, inverters[::-1]) if return_inverter else (image, labels)

    def augment_batch(self, images, labels, seeds=None):
        """The whole chain on a device-resident batch: images (B, H, W, 3) CUDA uint8, labels a list of B
        (n_i, 5) arrays -> ((B, img_height, img_width, 3) CUDA uint8 batch, list of B label arrays).  TWO launches for the pixels of
        the whole batch -- the photometric distortions (`ssdhip_image_program`, one program per image) and expansion + crop + flip +
        resize fused into one gather (`ssdhip_image_resize_gather_cv_u8`: the geometric ops only record index maps, nothing but the final
        batch is materialised) -- and no PCIe traffic besides the tap tables.  The random draws, the label arithmetic and the box
        filtering are the per-image chain's own code in the per-image chain's order: with the same NumPy random state the result
        equals calling the chain on image 0, 1, 2, ... (tests/test_image_ops.py).

        `seeds` (round 5): one integer per image.  Image i is then augmented exactly as `np.random.seed(seeds[i]); chain(image_i,
        labels_i)` would -- an independent stream per image is what lets the batch run in parallel: the host only makes each image's
        photometric draws, and ONE launch (`ssdhip_ssd_augment_decide`, a wave per image consuming that image's MT19937 stream as NumPy
        does) takes every other decision of the chain -- expansion, the crop search with its IoU validation, flip, interpolation mode --
        and does the label arithmetic.  The global NumPy generator is left untouched."""
        import torch
        if seeds is not None:
            return self._augment_batch_seeded(images, labels, seeds)
        if not (torch.is_tensor(images) and images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4 and images.shape[3] == 3):
            raise TypeError("augment_batch takes a (B, H, W, 3) CUDA uint8 batch")
        if len(labels) != images.shape[0]:
            raise ValueError("one label array per image")
        if os.environ.get("SSDHIP_AUG_HOST_STREAM", "0") != "1":
            # Round 6: the reference's own semantics -- ONE global np.random stream across the batch -- decided on the device: a single wave
            # walks the images in order on that stream (ssdhip_ssd_augment_decide_stream: photometric AND geometric decisions, label
            # arithmetic), the pixel launches are the seeded path's, and the generator state behind the last image goes back into
            # np.random.  None: a configuration / label layout the kernel does not cover -> the host loop below.
            done = self._augment_batch_seeded(images, labels, None)
            if done is not None:
                return done
        for t in (self.expand, self.random_crop, self.random_flip, self.resize):
            t.labels_format = self.labels_format
        h, w = int(images.shape[1]), int(images.shape[2])
        programs, lazies, out_labels = [], [], []
        for lab in labels:
            programs.append(self.photometric_distortions.draw())         # the draws of this image's photometric part, then its geometry
            img, lab = iop.GeoImage.of(h, w), np.asarray(lab)
            for transform in (self.expand, self.random_crop, self.random_flip, self.resize):
                img, lab = transform(img, lab)
            lazies.append(img)
            out_labels.append(lab)
        distorted = iop.run_batch(images, programs)
        return iop.gather_batch(distorted, lazies), out_labels

    def _seeded_params(self, h, w):
        """The chain's configuration as ssdhip_augment_params fields, or None when an op is not in the original-SSD configuration the
        kernel implements."""
        ex, cr = self.expand.expand, self.random_crop.random_crop
        eg, cg, bg = ex.patch_coord_generator, cr.patch_coord_generator, cr.bound_generator
        val, bf = cr.image_validator, cr.box_filter
        ok = (eg.must_match == 'h_w' and eg.scale_uniformly and ex.n_trials_max == 1 and ex.image_validator is None and ex.box_filter is None
              and not ex.clip_boxes and all(v is None for v in (eg.patch_ymin, eg.patch_xmin, eg.patch_height, eg.patch_width))
              and cg.must_match == 'h_w' and not cg.scale_uniformly
              and all(v is None for v in (cg.patch_ymin, cg.patch_xmin, cg.patch_height, cg.patch_width))
              and cr.clip_boxes and bg is not None and val is not None and val.overlap_criterion == 'iou' and val.n_boxes_min == 1
              and val.border_pixels == 'half' and bf is not None and bf.check_overlap and not bf.check_min_area and not bf.check_degenerate
              and bf.overlap_criterion == 'center_point' and len(bg.sample_space) <= 8
              and self.random_flip.dim == 'horizontal' and len(self.resize.interpolation_modes) <= 8
              and self.box_filter.check_degenerate and not self.box_filter.check_overlap and not self.box_filter.check_min_area
              and eg.min_scale >= 1.0 and cg.max_scale <= 1.0)
        if not ok:
            return None
        cdf = np.array(bg.weights, dtype=np.float64).cumsum()           # as np.random.choice(n, p=weights) builds it
        cdf /= cdf[-1]
        pad = lambda v, fill: list(v) + [fill] * (8 - len(v))
        return dict(img_height=int(h), img_width=int(w), expand_prob=float(ex.prob), expand_min_scale=float(eg.min_scale),
                    expand_max_scale=float(eg.max_scale), crop_prob=float(cr.prob), crop_min_scale=float(cg.min_scale),
                    crop_max_scale=float(cg.max_scale), crop_min_aspect_ratio=float(cg.min_aspect_ratio),
                    crop_max_aspect_ratio=float(cg.max_aspect_ratio), n_trials=int(max(1, cr.n_trials_max)), n_bounds=len(bg.sample_space),
                    bound_cdf=pad(cdf.tolist(), 2.0), bound_lower=pad([float(b[0]) for b in bg.sample_space], 0.0),
                    bound_upper=pad([float(b[1]) for b in bg.sample_space], 1.0), flip_prob=float(self.random_flip.prob),
                    n_modes=len(self.resize.interpolation_modes), interpolation_modes=pad([int(m) for m in self.resize.interpolation_modes], 0),
                    out_height=int(self.resize.height), out_width=int(self.resize.width), max_rounds=0)

    def _augment_batch_seeded(self, images, labels, seeds):
        import torch
        from .. import _native as nat
        if not (torch.is_tensor(images) and images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4 and images.shape[3] == 3):
            raise TypeError("augment_batch takes a (B, H, W, 3) CUDA uint8 batch")
        B, h, w = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
        stream_mode = seeds is None                      # one global generator for the whole batch (the reference's contract)
        if len(labels) != B or (not stream_mode and len(seeds) != B):
            raise ValueError("one label array and one seed per image")
        lf = self.labels_format
        cols = [lf['class_id'], lf['xmin'], lf['ymin'], lf['xmax'], lf['ymax']]
        arrs = [np.asarray(lab) for lab in labels]
        params = self._seeded_params(h, w)
        dtypes = {a.dtype for a in arrs}
        fast = (params is not None and len(dtypes) == 1 and next(iter(dtypes)) in (np.dtype(np.int64), np.dtype(np.float64))
                and all(a.ndim == 2 and a.shape[1] == 5 and a.shape[0] <= nat.AUG_MAX_BOXES for a in arrs) and sorted(cols) == [0, 1, 2, 3, 4])
        if stream_mode:
            photo = self._photo_params()
            state = np.random.get_state()
            if not fast or photo is None or state[0] != 'MT19937' or B == 0:
                return None
            lab_in = np.zeros((B, nat.AUG_MAX_BOXES, 5), dtype=np.float64)
            n_in = np.empty((B,), dtype=np.int32)
            for i, a in enumerate(arrs):
                n_in[i] = a.shape[0]
                if a.shape[0]:
                    lab_in[i, :a.shape[0]] = a[:, cols]
            mt = np.empty((625,), dtype=np.uint32)
            mt[:624], mt[624] = state[1], state[2]
            ops_dev, args_dev, geo_dev, fetch = nat.ssd_augment_decide_stream(params, photo, mt, lab_in, n_in, images.device)
            distorted = nat.image_program(images.contiguous(), ops_dev, args_dev, torch.uint8)     # both sequences end in 'to_u8'
            out, out_labels, mt_out = self._gather_from_decisions(images, distorted, geo_dev, fetch, arrs[0].dtype, cols, B, h, w)
            np.random.set_state((state[0], mt_out[:624], int(mt_out[624]), state[3], state[4]))    # the stream goes on behind the batch
            return out, out_labels
        saved = np.random.get_state()
        try:
            if not fast:                                 # anything the kernel does not cover: the per-image chain's own code, seeded per image
                programs, lazies, out_labels = [], [], []
                for t in (self.expand, self.random_crop, self.random_flip, self.resize):
                    t.labels_format = self.labels_format
                for lab, seed in zip(arrs, seeds):
                    np.random.seed(int(seed))
                    programs.append(self.photometric_distortions.draw())
                    img = iop.GeoImage.of(h, w)
                    for transform in (self.expand, self.random_crop, self.random_flip, self.resize):
                        img, lab = transform(img, lab)
                    lazies.append(img)
                    out_labels.append(lab)
                return iop.gather_batch(iop.run_batch(images, programs), lazies), out_labels
            # ---- host: the photometric draws of every image under its own seed; the generator state behind them goes to the device ----
            mt = np.empty((B, 625), dtype=np.uint32)
            lab_in = np.zeros((B, nat.AUG_MAX_BOXES, 5), dtype=np.float64)
            n_in = np.empty((B,), dtype=np.int32)
            programs = []
            for i in range(B):
                np.random.seed(int(seeds[i]))
                programs.append(self.photometric_distortions.draw())
                st = np.random.get_state()
                mt[i, :624] = st[1]
                mt[i, 624] = st[2]
                a = arrs[i]
                n_in[i] = a.shape[0]
                if a.shape[0]:
                    lab_in[i, :a.shape[0]] = a[:, cols]
        finally:
            np.random.set_state(saved)
        distorted = iop.run_batch(images, programs)
        geo_dev, fetch = nat.ssd_augment_decide(params, mt, lab_in, n_in, images.device)
        out, out_labels, mt_out = self._gather_from_decisions(images, distorted, geo_dev, fetch, arrs[0].dtype, cols, B, h, w)
        self.__dict__["_last_generator_states"] = mt_out     # (B, 625): where each image's stream stands behind its chain (tests)
        return out, out_labels

    def _photo_params(self):
        """SSDPhotometricDistortions' configuration as the fields of ssdhip_augment_photo, or None when it is not the original-SSD one
        (RandomChannelSwap must never fire: the device walk only consumes its firing draw)."""
        pd = self.photometric_distortions
        ops = (pd.random_brightness, pd.random_contrast, pd.random_saturation)
        if pd.random_channel_swap.prob != 0.0 or not all(0.0 <= o.prob <= 1.0 for o in ops + (pd.random_hue,)):
            return None
        return dict(prob=[float(o.prob) for o in ops] + [float(pd.random_hue.prob)],
                    lower=[float(o.lower) for o in ops] + [-float(pd.random_hue.max_delta)],
                    upper=[float(o.upper) for o in ops] + [float(pd.random_hue.max_delta)], swap_prob=0.0)

    def _gather_from_decisions(self, images, distorted, geo_dev, fetch, dt, cols, B, h, w):
        """The pixel half behind the device-side decisions: tap tables built on the device from `geo_dev`, ONE gather launch, then the one
        download of the call (labels, geometry, generator state).  Returns (batch, labels, generator state(s))."""
        from .. import _native as nat
        out_h, out_w = int(self.resize.height), int(self.resize.width)
        # tap tables wide enough for the true area filter of the largest possible source (an expanded, uncropped image)
        n_taps = max(8, int(np.ceil(float(self.expand.expand.patch_coord_generator.max_scale) * max(h / out_h, w / out_w))) + 2)
        inv = np.argsort(cols)
        if n_taps <= 64 and os.environ.get("SSDHIP_AUG_HOST_TAPS", "0") != "1":
            # ---- tap tables built on the device from the decisions, the gather launch behind them; the labels come back last (the only
            #      host synchronisation of the call) ------------------------------------------------------------------------------------
            plans, ix, wx, iy, wy = nat.augment_plans(geo_dev, h, w, out_h, out_w, n_taps)
            bg = self.__dict__.get("_bg_rows")
            if bg is None or bg[0] != (B, str(images.device)):
                row = np.array([int(v) for v in self.expand.expand.background], dtype=np.uint8)
                bg = ((B, str(images.device)), nat.to_device(np.repeat(row[None], B, 0), device=images.device))
                self.__dict__["_bg_rows"] = bg
            out = nat.image_resize_gather_cv_u8(distorted.contiguous(), out_h, out_w, plans, ix, wx, iy, wy, bg[1])
            geo, lab_out, n_out, mt_out = fetch()
            return out, [np.ascontiguousarray(lab_out[i, :int(n_out[i])][:, inv]).astype(dt) for i in range(B)], mt_out
        geo, lab_out, n_out, mt_out = fetch()
        # ---- the recorded geometry of every image -> one gather launch; the labels back in the caller's column order and dtype ----------
        lazies, out_labels = [], []
        for i in range(B):
            g = geo[i]
            img = iop.GeoImage.of(h, w)
            if g[0]:
                img = img.window(int(g[1]), int(g[2]), int(g[3]), int(g[4]), self.expand.expand.background)
            if g[5]:
                img = img.window(int(g[6]), int(g[7]), int(g[8]), int(g[9]), self.random_crop.random_crop.background)
            if g[10]:
                img = img[:, ::-1]
            lazies.append(img.resize(self.resize.height, self.resize.width, int(g[11])))
            out_labels.append(np.ascontiguousarray(lab_out[i, :int(n_out[i])][:, inv]).astype(dt))
        return iop.gather_batch(distorted, lazies), out_labels, mt_out
