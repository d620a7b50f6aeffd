"""Throughput of the image half of the augmentation on device-resident batches (csrc/ssdhip_image.hip): the photometric distortions of
the original-SSD chain as one launch per batch, and the resize to the network input in each interpolation mode.  GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd.data_generator import _image_ops as iop  # noqa: E402
from ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd import SSDPhotometricDistortions  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


rng = np.random.RandomState(0)
B = 32
voc = torch.from_numpy(rng.randint(0, 256, size=(B, 375, 500, 3)).astype(np.uint8)).cuda()
net = torch.from_numpy(rng.randint(0, 256, size=(B, 300, 300, 3)).astype(np.uint8)).cuda()
d = SSDPhotometricDistortions()
np.random.seed(0)
progs = [d.draw() for _ in range(B)]
out = {"batch": B}
us = timed(lambda: iop.run_batch(voc, progs))
out["photometric_375x500"] = {"us_per_batch_kernel_plus_program_upload": round(us, 1), "images_per_s": round(B / us * 1e6), "GBps_read_plus_write": round(2 * voc.numel() / us / 1e3, 1)}
us = timed(lambda: d.distort_batch(voc))
out["photometric_375x500_with_host_draws"] = {"us_per_batch": round(us, 1), "images_per_s": round(B / us * 1e6)}
for name, interp in (("nearest", 0), ("linear", 1), ("cubic", 2), ("area", 3), ("lanczos4", 4)):
    us = timed(lambda: iop.resize(voc, 300, 300, interp))
    out["resize_375x500_to_300x300_" + name] = {"us_per_batch_incl_tap_tables": round(us, 1), "images_per_s": round(B / us * 1e6)}
# the whole chain as a device pipeline (SSDDataAugmentation.augment_batch): wall clock incl. the host's draws, label arithmetic and tap tables
import time  # noqa: E402
from ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd import SSDDataAugmentation  # noqa: E402
aug = SSDDataAugmentation(img_height=300, img_width=300)
labels = []
for _ in range(B):
    n = rng.randint(1, 6)
    x0, y0 = rng.randint(0, 400, size=n), rng.randint(0, 280, size=n)
    labels.append(np.stack([rng.randint(1, 21, size=n), x0, y0, x0 + rng.randint(20, 100, size=n), y0 + rng.randint(20, 90, size=n)], axis=1))
np.random.seed(1)
aug.augment_batch(voc, labels)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    aug.augment_batch(voc, labels)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
out["augment_batch_375x500_to_300x300"] = {"ms_per_batch_wall": round(dt * 1e3, 2), "images_per_s": round(B / dt)}
print(json.dumps(out))
