#!/usr/bin/env python3
"""In-kernel phase timers of the decoder (needs tools/libssdhip_prof.so: tools/prof_build.sh).  GPU box only."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["SSDHIP_LIB"] = os.path.join(HERE, "libssdhip_prof.so")
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np      # noqa: E402
import torch            # noqa: E402

from ssd_keras_amd import _native as nat       # noqa: E402
from ssd_keras_amd import synthetic as syn     # noqa: E402
from tools.time_decode import anchors_var      # noqa: E402

NMS = ["select", "collect", "sort", "loadbox", "phaseA", "phaseB", "resolve"]
SCAN = ["tilecopy", "decode", "pass1", "atomics", "pass2"]
TOPK = ["offs", "select", "collect", "sort", "emit", "zero"]


def read(lib, reset=True):
    buf = (ctypes.c_ulonglong * 64)()
    assert lib.ssdhip_profile_read(buf, 1 if reset else 0) == 0
    return np.array(list(buf), dtype=np.float64)


def run(name, y, img=300):
    lib = nat.load()
    yd = torch.from_numpy(y).cuda()
    kw = dict(conf_thresh=0.01, iou_thresh=0.45, top_k=200, nms_cap=400, class_agnostic=False, semantics=nat.SEM_KERAS,
              coords="centroids", normalize_coords=True, img_height=img, img_width=img, border_pixels="half",
              out_dtype=nat.F32, out_rows=200)
    nat.decode(yd, **kw)
    torch.cuda.synchronize()
    read(lib)
    nat.decode(yd, **kw)
    torch.cuda.synchronize()
    g = read(lib)
    B, N, L = y.shape
    G = L - 13
    nblk = max(g[12], 1)
    print("== %s: nms blocks %d, consumed/blk %.0f, kept/blk %.1f, exact-select rounds %d (top bin alone overflows: %d)" % (
        name, g[12], g[13] / nblk, g[14] / nblk, g[15], g[11]))
    print("  nms  kcycles/block: " + "  ".join("%s %.1f" % (n, g[i] / nblk / 1e3) for i, n in enumerate(NMS)) + "   total %.1f" % (g[:7].sum() / nblk / 1e3))
    nscan = ((N + 255) // 256) * B
    print("  scan kcycles/block: " + "  ".join("%s %.2f" % (n, g[16 + i] / nscan / 1e3) for i, n in enumerate(SCAN)))
    print("  topk kcycles/block: " + "  ".join("%s %.1f" % (n, g[32 + i] / B / 1e3) for i, n in enumerate(TOPK)), flush=True)


if __name__ == "__main__":
    av = anchors_var(syn.SSD300_VOC)
    run("sparse_bias7", syn.make_y_pred(av, 32, 21, bias=7.0))
    run("dense_bias0", syn.make_y_pred(av, 32, 21, bias=0.0))
    run("dense_wild", syn.make_y_pred(av, 32, 21, bias=0.0, loc_sigma=300.0))
    if os.environ.get("MODEL", "1") == "1":          # what bench.py decodes: a random-init SSD300's own predictions
        from ssd_keras_amd.models.keras_ssd300 import ssd_300
        torch.manual_seed(1234)
        cfg = syn.SSD300_VOC
        model = ssd_300((300, 300, 3), 20, mode="training", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                        steps=cfg["steps"], offsets=cfg["offsets"]).cuda().to(memory_format=torch.channels_last).to(torch.bfloat16).eval()
        imgs = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(32, 300, 300, 3)).astype(np.float32)).cuda()
        with torch.no_grad():
            yp = model(imgs).float().cpu().numpy()
        run("random_init_model", yp)
        with torch.no_grad():                        # bench.py's value_tamed_heads workload: distinct, unsaturated confidences
            for head in model.conf_heads:
                head.weight.mul_(1e-2)
                head.bias.view(-1, 21)[:, 0] = 4.0
            for head in model.loc_heads:
                head.weight.mul_(1e-2)
            yt = model(imgs).float().cpu().numpy()
        run("tamed_heads_model", yt)
