"""Seeded cases for the patch sampling ops (data_generator/object_detection_2d_patch_sampling_ops.py and the box-level half of
data_augmentation_chain_original_ssd.py).  The same builder runs against the reference's classes (tests/golden/make_golden.py, in
the build container) and against the drop-in's classes (tests): `ns` is any object carrying the class names used below."""
import numpy as np

DEFAULT_FORMAT = {'class_id': 0, 'xmin': 1, 'ymin': 2, 'xmax': 3, 'ymax': 4}
ALT_FORMAT = {'class_id': 4, 'xmin': 0, 'ymin': 1, 'xmax': 2, 'ymax': 3}       # a non-default column order


def make_inputs(seed, n_boxes, float_labels=False, gray=False, size=(60, 80), fmt=DEFAULT_FORMAT):
    rng = np.random.RandomState(1000 + seed)
    h, w = size
    image = rng.randint(0, 256, size=(h, w) if gray else (h, w, 3)).astype(np.uint8)
    x0 = rng.uniform(0, 0.7 * w, size=n_boxes)
    y0 = rng.uniform(0, 0.7 * h, size=n_boxes)
    bw = rng.uniform(0.05 * w, 0.6 * w, size=n_boxes)
    bh = rng.uniform(0.05 * h, 0.6 * h, size=n_boxes)
    cols = np.zeros((n_boxes, 5), dtype=np.float64)
    cols[:, fmt['class_id']] = rng.randint(1, 21, size=n_boxes)
    cols[:, fmt['xmin']], cols[:, fmt['ymin']] = x0, y0
    cols[:, fmt['xmax']], cols[:, fmt['ymax']] = np.minimum(x0 + bw, w - 1), np.minimum(y0 + bh, h - 1)
    labels = cols if float_labels else np.round(cols).astype(np.int64)
    return image, labels


def _cases():
    cases = []
    for seed in range(36):                                   # the SSD crop: many seeds, 1-8 boxes, int and float labels
        cases.append(dict(op='ssd_random_crop', seed=seed, n_boxes=1 + seed % 8, float_labels=seed % 3 == 0, gray=seed % 7 == 0,
                          fmt='alt' if seed % 5 == 0 else 'default', inverter=seed % 4 == 0))
    for seed in range(8):
        cases.append(dict(op='ssd_expand', seed=100 + seed, n_boxes=3, float_labels=seed % 2 == 0, gray=seed == 5, inverter=seed % 3 == 0))
    # RandomPatchInf beyond the SSD configuration: other criteria, 'all' boxes, no bound generator, dependent width / height
    for i, (crit, nmin, match, bound_gen) in enumerate([('area', 1, 'h_ar', True), ('center_point', 2, 'w_ar', False),
                                                        ('iou', 'all', 'h_w', True), ('area', 'all', 'w_ar', False),
                                                        ('center_point', 1, 'h_ar', True), ('iou', 2, 'h_w', False)]):
        for seed in range(4):
            cases.append(dict(op='patch_inf', seed=200 + 10 * i + seed, n_boxes=2 + seed, float_labels=seed == 1, crit=crit, n_boxes_min=nmin,
                              must_match=match, bound_gen=bound_gen, n_trials_max=(3, 7, 1, 50)[seed], prob=(0.5, 0.857, 0.9, 1.0)[seed],
                              clip_boxes=seed != 2, box_filter=seed != 3, inverter=seed == 0))
    # RandomPatch: finite, may fail
    for i, can_fail in enumerate((False, True)):
        for seed in range(6):
            cases.append(dict(op='patch', seed=300 + 10 * i + seed, n_boxes=1 + seed % 3, float_labels=seed == 4, can_fail=can_fail,
                              crit=('iou', 'area', 'center_point')[seed % 3], bounds=((0.6, 1.0), (0.3, 1.0), (0.9, 1.0))[seed % 3],
                              n_trials_max=(1, 3, 6)[seed % 3], prob=(1.0, 0.7)[seed % 2], validator=seed != 5, inverter=seed % 2 == 1,
                              scale=((0.3, 1.0), (0.5, 2.0))[seed % 2]))
    for seed in range(4):
        cases.append(dict(op='max_crop_ar', seed=400 + seed, n_boxes=3, ar=(1.0, 0.5, 2.0, 1.7)[seed], validator=seed % 2 == 0, inverter=seed == 3))
        cases.append(dict(op='pad_ar', seed=410 + seed, n_boxes=3, ar=(1.0, 0.5, 2.0, 1.7)[seed], size=((60, 80), (80, 60))[seed % 2]))
    cases.append(dict(op='crop', seed=420, n_boxes=5, args=(5, 7, 11, 3), box_filter=True))
    cases.append(dict(op='crop', seed=421, n_boxes=5, args=(0, 0, 30, 30), box_filter=False, inverter=True))
    cases.append(dict(op='pad', seed=422, n_boxes=4, args=(3, 0, 9, 14), background=(10, 20, 30)))
    cases.append(dict(op='pad', seed=423, n_boxes=4, args=(3, 2, 1, 0), background=(7, 8, 9), gray=True))
    for i, (top, left, ph, pw) in enumerate([(-10, -12, 90, 120), (-10, 20, 40, 30), (15, -8, 30, 50), (10, 20, 100, 100), (0, 0, 60, 80),
                                             (59, 79, 5, 5)]):
        cases.append(dict(op='crop_pad', seed=430 + i, n_boxes=6, args=(top, left, ph, pw), clip_boxes=i % 2 == 0, box_filter=i % 3 != 2,
                          float_labels=i == 3, background=(1, 2, 3), inverter=i == 1))
    # no validator at all (any patch of an acceptable shape is cut), a non-default label layout, and validators that draw
    # their own random bounds on every call (the trials of a round then cannot be validated in one go)
    for seed in range(4):
        cases.append(dict(op='patch_inf', seed=500 + seed, n_boxes=3, float_labels=seed == 2, crit='iou', n_boxes_min=1, must_match='h_w',
                          bound_gen=False, n_trials_max=5, prob=0.8, clip_boxes=True, box_filter=seed % 2 == 0, inverter=seed == 1,
                          validator=False, fmt='alt' if seed == 3 else 'default'))
    for seed in range(6):
        cases.append(dict(op='patch_inf', seed=510 + seed, n_boxes=2 + seed % 3, float_labels=seed == 4, crit=('iou', 'area')[seed % 2],
                          n_boxes_min=1, must_match='h_w', bound_gen=False, n_trials_max=(4, 9)[seed % 2], prob=0.9, clip_boxes=True,
                          box_filter=True, inverter=seed == 5, random_bounds=True, fmt='alt' if seed == 2 else 'default'))
    for seed in range(4):
        cases.append(dict(op='patch', seed=520 + seed, n_boxes=2, float_labels=False, can_fail=seed % 2 == 0, crit='iou', bounds=None,
                          n_trials_max=4, prob=1.0, validator=True, inverter=seed == 3, scale=(0.3, 1.0), random_bounds=True))
    return cases


CASES = _cases()


def build(ns, case):
    fmt = ALT_FORMAT if case.get('fmt') == 'alt' else DEFAULT_FORMAT
    op = case['op']
    if op == 'ssd_random_crop':
        return ns.SSDRandomCrop(labels_format=fmt)
    if op == 'ssd_expand':
        return ns.SSDExpand(labels_format=fmt)
    if op == 'patch_inf':
        gen = ns.PatchCoordinateGenerator(must_match=case['must_match'], min_scale=0.3, max_scale=1.0, min_aspect_ratio=0.5, max_aspect_ratio=2.0)
        box_filter = ns.BoxFilter(check_overlap=True, check_min_area=True, check_degenerate=True, overlap_criterion='area',
                                  overlap_bounds=(0.4, 1.0), min_area=20, labels_format=fmt) if case['box_filter'] else None
        vbounds = ns.BoundGenerator(sample_space=((0.05, None), (0.2, None), (0.4, 0.95))) if case.get('random_bounds') else (0.3, 1.0)
        validator = ns.ImageValidator(overlap_criterion=case['crit'], bounds=vbounds, n_boxes_min=case['n_boxes_min'], labels_format=fmt)
        if not case.get('validator', True):
            validator = None
        bounds = ns.BoundGenerator(sample_space=((0.1, None), (0.3, 0.9), (None, None)), weights=(0.5, 0.25, 0.25)) if case['bound_gen'] else None
        return ns.RandomPatchInf(gen, box_filter=box_filter, image_validator=validator, bound_generator=bounds,
                                 n_trials_max=case['n_trials_max'], clip_boxes=case['clip_boxes'], prob=case['prob'], background=(9, 8, 7),
                                 labels_format=fmt)
    if op == 'patch':
        gen = ns.PatchCoordinateGenerator(must_match='h_w', min_scale=case['scale'][0], max_scale=case['scale'][1])
        vbounds = ns.BoundGenerator(sample_space=((0.05, None), (0.2, None), (0.4, 0.95))) if case.get('random_bounds') else case['bounds']
        validator = ns.ImageValidator(overlap_criterion=case['crit'], bounds=vbounds, n_boxes_min=1) if case['validator'] else None
        box_filter = ns.BoxFilter(check_overlap=True, check_min_area=False, check_degenerate=True, overlap_criterion='center_point')
        return ns.RandomPatch(gen, box_filter=box_filter, image_validator=validator, n_trials_max=case['n_trials_max'], clip_boxes=True,
                              prob=case['prob'], background=(50, 60, 70), can_fail=case['can_fail'])
    if op == 'max_crop_ar':
        validator = ns.ImageValidator(overlap_criterion='center_point', n_boxes_min=1) if case['validator'] else None
        return ns.RandomMaxCropFixedAR(case['ar'], box_filter=ns.BoxFilter(overlap_criterion='center_point'), image_validator=validator,
                                       n_trials_max=3, clip_boxes=True)
    if op == 'pad_ar':
        return ns.RandomPadFixedAR(case['ar'], background=(4, 5, 6))
    if op == 'crop':
        bf = ns.BoxFilter(check_overlap=True, check_min_area=True, check_degenerate=True, overlap_criterion='iou', overlap_bounds=(0.05, 1.0),
                          min_area=4) if case['box_filter'] else None
        return ns.Crop(*case['args'], clip_boxes=True, box_filter=bf)
    if op == 'pad':
        return ns.Pad(*case['args'], background=case['background'])
    if op == 'crop_pad':
        bf = ns.BoxFilter(check_overlap=True, check_min_area=False, check_degenerate=False, overlap_criterion='center_point') if case['box_filter'] else None
        return ns.CropPad(*case['args'], clip_boxes=case['clip_boxes'], box_filter=bf, background=case['background'])
    raise KeyError(op)


def run(ns, case):
    """Seeds NumPy's global stream, runs the op once, returns {name: array}: the patch, its labels, the inverter applied to a fixed
    prediction array, and three draws from the stream after the call (pins how much of it the op consumed)."""
    fmt = ALT_FORMAT if case.get('fmt') == 'alt' else DEFAULT_FORMAT
    image, labels = make_inputs(case['seed'], case['n_boxes'], case.get('float_labels', False), case.get('gray', False),
                                case.get('size', (60, 80)), fmt)
    op = build(ns, case)
    want_inv = bool(case.get('inverter', False))
    np.random.seed(case['seed'])
    with np.errstate(divide='ignore', invalid='ignore'):
        res = op(image, labels, return_inverter=True) if want_inv else op(image, labels)
    tail = np.random.uniform(size=3)
    out = {'tail': tail}
    img, lab = res[0], res[1]
    out['image_none'] = np.array(img is None)
    out['labels_none'] = np.array(lab is None)
    if img is not None:
        out['image'] = np.asarray(img)
    if lab is not None:
        out['labels'] = np.asarray(lab)
    if want_inv:
        inv = res[2]
        out['inverter_none'] = np.array(inv is None)
        if inv is not None:
            pred = np.arange(18, dtype=np.float64).reshape(3, 6) * 1.5
            out['inverted'] = np.asarray(inv(pred))
    return out
