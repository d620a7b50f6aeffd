"""cProfile of SSDDataAugmentation.augment_batch on a device-resident batch (where do the host milliseconds go).  GPU box."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd import SSDDataAugmentation  # noqa: E402

B = int(os.environ.get("B", "32"))
rng = np.random.RandomState(0)
voc = torch.from_numpy(rng.randint(0, 256, size=(B, 375, 500, 3)).astype(np.uint8)).cuda()
aug = SSDDataAugmentation(img_height=300, img_width=300)
labels = []
for _ in range(B):
    n = rng.randint(1, 6)
    x0, y0 = rng.randint(0, 400, size=n), rng.randint(0, 280, size=n)
    labels.append(np.stack([rng.randint(1, 21, size=n), x0, y0, x0 + rng.randint(20, 100, size=n), y0 + rng.randint(20, 90, size=n)], axis=1))
np.random.seed(1)
SEEDED = os.environ.get("SEEDED", "1") == "1"
seeds = lambda: (np.random.randint(0, 2 ** 31 - 1, size=B) if SEEDED else None)
aug.augment_batch(voc, labels, seeds=seeds())
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    aug.augment_batch(voc, labels, seeds=seeds())
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print("augment_batch (%s): %.2f ms per batch of %d = %.0f img/s" % ("one seed per image, decisions on the device" if SEEDED else "global stream, per-image host loop", dt * 1e3, B, B / dt))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    aug.augment_batch(voc, labels, seeds=seeds())
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
