"""Encoder timing (GPU box): SSD300 / batch 32 / 1-8 GT boxes per image, events over 50 calls; also g = 16 and SSD512."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder  # noqa: E402

dev = torch.device("cuda:0")
for name, cfg, B, mx, img in (("ssd300_g8", syn.SSD300_VOC, 32, 8, 300), ("ssd300_g16", syn.SSD300_VOC, 32, 16, 300), ("ssd512_g8", syn.SSD512_COCO, 16, 8, 512)):
    enc = SSDInputEncoder(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
    gt = syn.make_ground_truth(B, cfg["n_classes"], img, img, max_boxes=mx, seed=7)
    enc.encode_to_device(gt, device=dev)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        y, _, _ = enc.encode_to_device(gt, device=dev)
    b.record()
    b.synchronize()
    ms = a.elapsed_time(b) / 50
    nbytes = y.numel() * 4
    gt_np, off_np, max_g = enc._pack_ground_truth(gt)
    gt_d, off_d = torch.from_numpy(gt_np).to(dev), torch.from_numpy(off_np).to(dev)
    enc.encode_packed(gt_d, off_d, gt_np.shape[0], max_g, B)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        enc.encode_packed(gt_d, off_d, gt_np.shape[0], max_g, B)
    b.record()
    b.synchronize()
    km = a.elapsed_time(b) / 50
    print(json.dumps({"case": name, "ms_per_batch_with_host": round(ms, 4), "ms_kernels": round(km, 4),
                      "GBps_on_the_f32_write": round(nbytes / km / 1e6, 1)}), flush=True)
