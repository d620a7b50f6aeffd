"""What a user pays for `model.precise()`: the call itself, the FIRST forward (filter packing, the divisor calibration, the side-stream
pick) and the second one.  `bench` as argv[1]: with torch.backends.cudnn.benchmark on.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ssd_keras_amd import synthetic as syn
from ssd_keras_amd.models.keras_ssd300 import ssd_300
if len(sys.argv) > 1 and sys.argv[1] == "bench":
    torch.backends.cudnn.benchmark = True
cfg = syn.SSD300_VOC
torch.manual_seed(1234)
model = ssd_300((300, 300, 3), cfg["n_classes"], mode="inference", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                steps=cfg["steps"], offsets=cfg["offsets"], confidence_thresh=0.01, iou_threshold=0.45, top_k=200,
                nms_max_output_size=400).cuda().to(memory_format=torch.channels_last).eval()
images = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(32, 300, 300, 3)).astype(np.float32)).cuda()
torch.cuda.synchronize(); t = time.perf_counter()
model.precise()
torch.cuda.synchronize(); t1 = time.perf_counter()
with torch.no_grad():
    model(images)
torch.cuda.synchronize(); t2 = time.perf_counter()
with torch.no_grad():
    model(images)
torch.cuda.synchronize(); t3 = time.perf_counter()
print("precise() %.2f s, first call %.2f s, second call %.4f s" % (t1 - t, t2 - t1, t3 - t2))
