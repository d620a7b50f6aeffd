"""Seeded cases for `eval_utils/coco_utils.predict_all_to_json`, shared by tests/golden/make_golden.py (run against the REAL reference
module) and tests/test_coco_utils.py (run against the drop-in).  `run(mod, case, tmpdir)` drives `mod.predict_all_to_json` with a
stand-in data generator that honours the reference generator's contract for the arguments that function passes
(data_generator/object_detection_2d_data_generator.py:1049-1092: per image every transformation in order, `return_inverter=True`
where the transformation has that parameter, the inverters reversed) and a stand-in model whose `predict` replays stored arrays."""
import inspect
import json
import os

import numpy as np

CASES = [
    dict(name="inference_resize", mode="resize", model_mode="inference", n_images=7, batch=3, seed=1, H=48, W=64),
    dict(name="inference_pad", mode="pad", model_mode="inference", n_images=5, batch=2, seed=2, H=60, W=40),
    dict(name="inference_fast_resize_grey", mode="resize", model_mode="inference_fast", n_images=4, batch=4, seed=3, H=32, W=32, grey=True),
    dict(name="training_resize", mode="resize", model_mode="training", n_images=5, batch=2, seed=4, H=96, W=128),
    dict(name="training_pad", mode="pad", model_mode="training", n_images=3, batch=3, seed=5, H=96, W=128),
]
CLASSES_TO_CATS = {1: 1, 2: 2, 3: 4, 4: 7, 5: 90}


def make_images(case):
    rng = np.random.RandomState(100 + case["seed"])
    images = []
    for i in range(case["n_images"]):
        h, w = int(rng.randint(20, 90)), int(rng.randint(20, 90))
        if case.get("grey") and i % 2 == 0:
            images.append(rng.randint(0, 256, size=(h, w)).astype(np.uint8))                 # ConvertTo3Channels has work to do
        elif case.get("grey"):
            images.append(rng.randint(0, 256, size=(h, w, 4)).astype(np.uint8))
        else:
            images.append(rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8))
    return images


def make_predictions(case, n_batches_sizes):
    """What `model.predict` returns per batch: decoded, zero-padded `(b, 12, 6)` rows for the inference modes; raw `(b, N, C + 12)`
    predictions over the TINY anchor grid for 'training' (decoded by the module under test)."""
    rng = np.random.RandomState(200 + case["seed"])
    out = []
    if case["model_mode"] != "training":
        for b in n_batches_sizes:
            y = np.zeros((b, 12, 6), dtype=np.float32)
            for i in range(b):
                k = int(rng.randint(0, 9))                                                     # 0 rows happens
                y[i, :k, 0] = rng.randint(1, 6, size=k)
                y[i, :k, 1] = rng.uniform(0.01, 1.0, size=k)
                x0, y0 = rng.uniform(-5, case["W"] * 0.7, size=k), rng.uniform(-5, case["H"] * 0.7, size=k)
                y[i, :k, 2], y[i, :k, 3] = x0, y0
                y[i, :k, 4], y[i, :k, 5] = x0 + rng.uniform(1, case["W"] * 0.5, size=k), y0 + rng.uniform(1, case["H"] * 0.5, size=k)
            out.append(y)
        return out
    from ssd_keras_amd import synthetic as syn
    from oracle import np_oracle as orc
    cfg = dict(syn.TINY, img_height=case["H"], img_width=case["W"])
    anchors = orc.EncoderOracle(**cfg).generate_encoding_template(1)[0, :, -8:]
    for j, b in enumerate(n_batches_sizes):
        out.append(syn.make_y_pred(anchors, b, 6, bias=5.0, seed=300 + 10 * case["seed"] + j, loc_sigma=0.3))
    return out


class Generator:
    def __init__(self, images):
        self.images = images
        self.image_ids = [1000 + 7 * i for i in range(len(images))]
        self.labels = None
        self.calls = []

    def get_dataset_size(self):
        return len(self.images)

    def generate(self, batch_size=32, shuffle=True, transformations=[], label_encoder=None, returns={'processed_images', 'encoded_labels'},
                 keep_images_without_gt=False, degenerate_box_handling='remove'):
        self.calls.append(dict(batch_size=batch_size, shuffle=shuffle, label_encoder=label_encoder, returns=sorted(returns),            # (sorted: a set's repr depends on the hash seed)
                               keep_images_without_gt=keep_images_without_gt,
                               transformations=[type(t).__name__ for t in transformations]))
        current = 0
        while True:
            if current >= len(self.images):
                current = 0
            batch_X = [np.copy(im) for im in self.images[current:current + batch_size]]
            ids = self.image_ids[current:current + batch_size]
            current += batch_size
            inverse = []
            for i in range(len(batch_X)):
                inv = []
                for transform in transformations:
                    if 'return_inverter' in inspect.signature(transform).parameters:
                        batch_X[i], inverter = transform(batch_X[i], return_inverter=True)
                        inv.append(inverter)
                    else:
                        batch_X[i] = transform(batch_X[i])
                inverse.append(inv[::-1])
            yield np.array(batch_X), ids, inverse              # (the order of the reference's return tuple for these three names)


class Model:
    def __init__(self, replies):
        self.replies = list(replies)
        self.seen = []

    def predict(self, batch_X):
        self.seen.append(np.asarray(batch_X).shape)
        return self.replies[len(self.seen) - 1]


def run(mod, case, tmpdir, seed_np=True):
    images = make_images(case)
    sizes = [min(case["batch"], case["n_images"] - s) for s in range(0, case["n_images"], case["batch"])]
    replies = make_predictions(case, sizes)
    gen, model = Generator(images), Model(replies)
    out_file = os.path.join(tmpdir, case["name"] + ".json")
    if seed_np:
        np.random.seed(case["seed"])                            # RandomPadFixedAR draws from the global stream
    try:
        mod.predict_all_to_json(out_file, model, case["H"], case["W"], CLASSES_TO_CATS, gen, case["batch"], data_generator_mode=case["mode"],
                                model_mode=case["model_mode"], confidence_thresh=0.2, iou_threshold=0.45, top_k=20)
    except TypeError as exc:                                    # the reference's 'pad' mode: see tests/test_coco_utils.py
        return dict(json="", results=None, error="TypeError: %s" % exc, generate_call=gen.calls, batches_seen=model.seen)
    with open(out_file) as f:
        text = f.read()
    return dict(json=text, results=json.loads(text), generate_call=gen.calls, batches_seen=model.seen)
