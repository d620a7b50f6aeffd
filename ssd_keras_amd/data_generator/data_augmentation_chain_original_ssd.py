"""Drop-in for the box-level half of the reference's data_generator/data_augmentation_chain_original_ssd.py (SURVEY section 8f row
4): `SSDRandomCrop` :29-101 and `SSDExpand` :103-162 -- the random crops (Caffe SSD `batch_sampler`) and the random expansion
(`train_transform_param`) of the original SSD training pipeline.  Both are configurations of the patch sampling ops; the IoU
validation of the candidate crops runs on the GPU, one launch per sampling round (object_detection_2d_patch_sampling_ops.py).

Not here: `SSDPhotometricDistortions` and the `SSDDataAugmentation` chain that strings all of them together with a resize -- they
are OpenCV colour-space and interpolation calls (SURVEY section 8f row 4's image half), not part of the box path.
"""
from __future__ import annotations

from .object_detection_2d_image_boxes_validation_utils import BoundGenerator, BoxFilter, ImageValidator
from .object_detection_2d_patch_sampling_ops import PatchCoordinateGenerator, RandomPatch, RandomPatchInf

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
