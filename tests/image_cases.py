"""Seeded cases for the image half of the augmentation (data_generator/object_detection_2d_photometric_ops.py, the resize / flip ops
of object_detection_2d_geometric_ops.py, SSDPhotometricDistortions and SSDDataAugmentation).  The same builder runs against the
reference's classes (tests/golden/make_golden.py gen_image_ops, in the build container, with a cv2 stub built on oracle/np_image.py)
and against the drop-in's classes (tests/test_image_ops.py): `ns` is any object carrying the class names used below."""
import numpy as np

from tests.patch_cases import DEFAULT_FORMAT, make_inputs


def make_image(seed, size=(20, 24), dtype="uint8", smooth=False):
    rng = np.random.RandomState(5000 + seed)
    h, w = size
    if smooth:                                               # gradients + a few flat patches: grey pixels, saturated pixels, ties of max(r, g, b)
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 255 // max(h + w - 2, 1))], axis=-1)
        img[:3, :3] = 0
        img[-3:, -3:] = 255
        img[:3, -3:] = (200, 200, 200)
        img[-3:, :3] = (255, 0, 0)
        img = img.astype(np.uint8)
    else:
        img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    if dtype == "float32":
        return (img.astype(np.float32) + rng.uniform(-0.4, 0.4, size=img.shape).astype(np.float32)).clip(0, 255).astype(np.float32)
    return img


def _cases():
    cases = []
    k = 0
    for dtype in ("uint8", "float32"):
        for smooth in (False, True):
            cases.append(dict(op="cvt", seed=k, dtype=dtype, smooth=smooth, current="RGB", to="HSV")); k += 1
            cases.append(dict(op="cvt", seed=k, dtype=dtype, smooth=smooth, current="RGB", to="GRAY", keep_3ch=True)); k += 1
            cases.append(dict(op="cvt", seed=k, dtype=dtype, smooth=smooth, current="RGB", to="GRAY", keep_3ch=False)); k += 1
            cases.append(dict(op="cvt_roundtrip", seed=k, dtype=dtype, smooth=smooth)); k += 1
    for dtype in ("uint8", "float32"):
        cases.append(dict(op="dtype", seed=k, dtype=dtype, to="uint8")); k += 1
        cases.append(dict(op="dtype", seed=k, dtype=dtype, to="float32")); k += 1
        for val in (-32.0, 17.25, 300.0):
            cases.append(dict(op="brightness", seed=k, dtype=dtype, value=val)); k += 1
        for val in (0.5, 1.37):
            cases.append(dict(op="contrast", seed=k, dtype=dtype, value=val)); k += 1
            cases.append(dict(op="saturation", seed=k, dtype=dtype, value=val)); k += 1
        for val in (-18.0, 7.5, 179.0):          # (floats: NumPy 2 refuses `uint8 array + negative Python int`, and wraps a positive one)
            cases.append(dict(op="hue", seed=k, dtype=dtype, value=val)); k += 1
    cases.append(dict(op="to3", seed=k, channels=1)); k += 1
    cases.append(dict(op="to3", seed=k, channels=4)); k += 1
    cases.append(dict(op="to3", seed=k, channels=0)); k += 1
    cases.append(dict(op="swap", seed=k, order=(2, 0, 1))); k += 1
    for smooth in (False, True):
        cases.append(dict(op="histeq", seed=k, smooth=smooth)); k += 1
    cases.append(dict(op="histeq", seed=k, constant=True)); k += 1
    for name in ("RandomHue", "RandomSaturation", "RandomBrightness", "RandomContrast", "RandomChannelSwap", "RandomHistogramEqualization"):
        for rep in range(3):
            cases.append(dict(op="random", cls=name, seed=k, prob=(0.5, 1.0, 0.0)[rep])); k += 1
    for seed in range(24):
        cases.append(dict(op="ssd_photometric", seed=600 + seed, smooth=seed % 4 == 0, size=((20, 24), (17, 31))[seed % 2]))
    for interp in range(5):
        for (size, out) in (((20, 24), (30, 36)), ((40, 48), (15, 13)), ((23, 37), (23, 50)), ((30, 30), (10, 10))):
            cases.append(dict(op="resize", seed=k, interp=interp, size=size, out=out, labels=interp % 2 == 0, inverter=interp == 1,
                              box_filter=interp == 4)); k += 1
    for seed in range(6):
        cases.append(dict(op="resize_random", seed=700 + seed, size=(33, 45), out=(20, 20)))
    for dim in ("horizontal", "vertical"):
        cases.append(dict(op="flip", seed=k, dim=dim, labels=True)); k += 1
        cases.append(dict(op="flip", seed=k, dim=dim, labels=False)); k += 1
    for seed in range(4):
        cases.append(dict(op="random_flip", seed=720 + seed, prob=(0.5, 1.0)[seed % 2]))
    for seed in range(10):
        cases.append(dict(op="ssd_augmentation", seed=800 + seed, n_boxes=1 + seed % 5, inverter=seed % 3 == 0, out=(30, 30)))
    return cases


CASES = _cases()


def _probe():
    return np.array(np.random.uniform(0, 1, size=3))      # pins how many random numbers the op consumed


def run(ns, case):
    """Run one case against the classes of `ns`; returns a dict of arrays."""
    op = case["op"]
    out = {}
    np.random.seed(case["seed"])
    if op in ("cvt", "cvt_roundtrip"):
        img = make_image(case["seed"], dtype=case["dtype"], smooth=case["smooth"])
        if op == "cvt":
            res = ns.ConvertColor(current=case["current"], to=case["to"], keep_3ch=case.get("keep_3ch", True))(img)
        else:
            hsv = ns.ConvertColor(current="RGB", to="HSV")(img)
            res = ns.ConvertColor(current="HSV", to="RGB")(hsv)
            out["mid"] = np.asarray(hsv)
        out["image"] = np.asarray(res)
    elif op == "dtype":
        img = make_image(case["seed"], dtype=case["dtype"])
        out["image"] = np.asarray(ns.ConvertDataType(to=case["to"])(img))
    elif op in ("brightness", "contrast", "saturation", "hue"):
        img = make_image(case["seed"], dtype=case["dtype"])
        if op in ("saturation", "hue") and case["dtype"] == "uint8":
            img[..., 0] = img[..., 0] % 180                    # a plausible 8-bit HSV image
        cls = {"brightness": ns.Brightness, "contrast": ns.Contrast, "saturation": ns.Saturation, "hue": ns.Hue}[op]
        res, labels = cls(case["value"])(img, np.zeros((1, 5)))
        out["image"] = np.asarray(res)
    elif op == "to3":
        rng = np.random.RandomState(case["seed"])
        c = case["channels"]
        img = rng.randint(0, 256, size=(9, 11) if c == 0 else (9, 11, c)).astype(np.uint8)
        out["image"] = np.asarray(ns.ConvertTo3Channels()(img))
    elif op == "swap":
        out["image"] = np.asarray(ns.ChannelSwap(order=case["order"])(make_image(case["seed"])))
    elif op == "histeq":
        img = make_image(case["seed"], smooth=case.get("smooth", False))
        if case.get("constant"):
            img[..., 2] = 77
        out["image"] = np.asarray(ns.HistogramEqualization()(img))
    elif op == "random":
        dtype = "uint8" if case["cls"] in ("RandomChannelSwap", "RandomHistogramEqualization") else "float32"
        img = make_image(case["seed"], dtype=dtype)
        res = getattr(ns, case["cls"])(prob=case["prob"])(img)
        out["image"] = np.asarray(res)
        out["probe"] = _probe()
    elif op == "ssd_photometric":
        img = make_image(case["seed"], size=case["size"], smooth=case["smooth"])
        labels = np.array([[1, 2, 3, 10, 12]])
        res, lab = ns.SSDPhotometricDistortions()(img, labels)
        out["image"], out["labels"], out["probe"] = np.asarray(res), np.asarray(lab), _probe()
    elif op in ("resize", "resize_random"):
        img, labels = make_inputs(case["seed"], 4, size=case["size"])
        bf = ns.BoxFilter(check_overlap=False, check_min_area=False, check_degenerate=True) if case.get("box_filter") else None
        if op == "resize":
            t = ns.Resize(height=case["out"][0], width=case["out"][1], interpolation_mode=case["interp"], box_filter=bf)
        else:
            t = ns.ResizeRandomInterp(height=case["out"][0], width=case["out"][1])
        if op == "resize" and not case["labels"]:
            res = t(img, None, return_inverter=case["inverter"])
            if case["inverter"]:
                res, inv = res
                out["inverted"] = inv(np.array([[1, 0.9, 3.0, 4.0, 9.0, 8.0]]))
            out["image"] = np.asarray(res)
        else:
            res = t(img, labels, return_inverter=case.get("inverter", False))
            out["image"], out["labels"] = np.asarray(res[0]), np.asarray(res[1])
            if case.get("inverter"):
                out["inverted"] = res[2](np.array([[1, 0.9, 3.0, 4.0, 9.0, 8.0]]))
        out["probe"] = _probe()
    elif op == "flip":
        img, labels = make_inputs(case["seed"], 3, size=(14, 18))
        if case["labels"]:
            res, lab = ns.Flip(dim=case["dim"])(img, labels)
            out["labels"] = np.asarray(lab)
        else:
            res = ns.Flip(dim=case["dim"])(img)
        out["image"] = np.ascontiguousarray(res)
    elif op == "random_flip":
        img, labels = make_inputs(case["seed"], 3, size=(14, 18))
        res, lab = ns.RandomFlip(dim="horizontal", prob=case["prob"])(img, labels)
        out["image"], out["labels"], out["probe"] = np.ascontiguousarray(res), np.asarray(lab), _probe()
    elif op == "ssd_augmentation":
        img, labels = make_inputs(case["seed"], case["n_boxes"], size=(40, 52))
        aug = ns.SSDDataAugmentation(img_height=case["out"][0], img_width=case["out"][1])
        res = aug(img, labels, return_inverter=case["inverter"])
        out["image"], out["labels"] = np.ascontiguousarray(res[0]), np.asarray(res[1])
        if case["inverter"]:
            pred = np.array([[1, 0.9, 3.0, 4.0, 19.0, 18.0], [2, 0.8, 1.0, 2.0, 25.0, 28.0]])
            for inv in res[2]:
                pred = inv(pred)
            out["inverted"] = np.asarray(pred)
        out["probe"] = _probe()
    else:
        raise ValueError(op)
    return out
