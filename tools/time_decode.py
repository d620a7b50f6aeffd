#!/usr/bin/env python3
"""Stage timings of the HIP decoder on the synthetic regimes of SURVEY 8d (GPU box)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_keras_amd import _native as nat          # noqa: E402
from ssd_keras_amd import anchor_math              # noqa: E402
from ssd_keras_amd import synthetic as syn         # noqa: E402


def anchors_var(cfg):
    L = len(cfg["predictor_sizes"])
    scales = cfg.get("scales")
    ars = cfg.get("aspect_ratios_per_layer") or [cfg["aspect_ratios_global"]] * L
    steps = cfg.get("steps") or [None] * L
    offs = cfg.get("offsets") or [None] * L
    a = np.concatenate([anchor_math.layer_anchor_boxes(cfg["img_height"], cfg["img_width"], cfg["predictor_sizes"][i], ars[i],
                                                       scales[i], scales[i + 1], True, steps[i], offs[i], False, "centroids",
                                                       True).reshape(-1, 4) for i in range(L)])
    return np.concatenate([a, np.zeros_like(a) + np.array(cfg["variances"])], axis=1)


def ev_ms(fn, reps=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps


def run(name, y, keras=True, top_k=200, cap=400, img=300):
    yd = torch.from_numpy(y).cuda()
    kw = dict(conf_thresh=0.01, iou_thresh=0.45, top_k=top_k, nms_cap=cap if keras else 0, class_agnostic=False,
              semantics=nat.SEM_KERAS if keras else nat.SEM_NUMPY, coords="centroids", normalize_coords=True, img_height=img,
              img_width=img, border_pixels="half", out_dtype=nat.F32 if keras else nat.F64, out_rows=top_k)
    outs = nat.decode(yd, **kw)
    torch.cuda.synchronize()
    res = {"case": name, "B": y.shape[0], "N": y.shape[1], "C": y.shape[2] - 12,
           "cand_per_img": float((y[:, :, 1:-12] > 0.01).sum() / y.shape[0]), "rows": float(outs[1].float().mean().item())}
    for nm, mask in (("scan", 1), ("nms", 2), ("topk", 4), ("all", 7)):
        res[nm + "_us"] = round(1e3 * ev_ms(lambda: nat.decode(yd, stages=mask, outputs=outs, **kw)), 2)
    print(json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    av = anchors_var(syn.SSD300_VOC)
    B = int(os.environ.get("B", "32"))
    results = []
    results.append(run("ssd300_sparse_bias7", syn.make_y_pred(av, B, 21, bias=7.0)))
    results.append(run("ssd300_mid_bias4", syn.make_y_pred(av, B, 21, bias=4.0)))
    results.append(run("ssd300_dense_bias0", syn.make_y_pred(av, B, 21, bias=0.0)))
    results.append(run("ssd300_dense_bias0_numpy_sem", syn.make_y_pred(av, B, 21, bias=0.0), keras=False))
    wild = syn.make_y_pred(av, B, 21, bias=0.0, loc_sigma=300.0)       # random-init-weights-like: exploding offsets
    results.append(run("ssd300_dense_wild_offsets", wild))
    if os.environ.get("MODEL", "1") == "1":
        from ssd_keras_amd.models.keras_ssd300 import ssd_300
        torch.manual_seed(1234)
        cfg = syn.SSD300_VOC
        model = ssd_300((300, 300, 3), 20, mode="training", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                        steps=cfg["steps"], offsets=cfg["offsets"]).cuda().to(memory_format=torch.channels_last).to(torch.bfloat16).eval()
        imgs = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).cuda()
        with torch.no_grad():
            yp = model(imgs).float().cpu().numpy()
        np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pred_rand_b0.npy"), yp[:2, :, :25])
        results.append(run("ssd300_random_init_bf16_model", yp))
        with torch.no_grad():                        # bench.py's value_tamed_heads workload: distinct, unsaturated confidences
            for head in model.conf_heads:
                head.weight.mul_(1e-2)
                head.bias.view(-1, 21)[:, 0] = 4.0
            for head in model.loc_heads:
                head.weight.mul_(1e-2)
            yt = model(imgs).float().cpu().numpy()
        results.append(run("ssd300_tamed_heads_bf16_model", yt))
    if os.environ.get("S512", "1") == "1":
        av5 = anchors_var(syn.SSD512_COCO)
        results.append(run("ssd512_sparse_bias7", syn.make_y_pred(av5, 16, 81, bias=7.0), img=512))
        results.append(run("ssd512_dense_bias0", syn.make_y_pred(av5, 16, 81, bias=0.0), img=512))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", os.environ.get("TDOUT", "time_decode.json"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(results, open(out, "w"), indent=1)
