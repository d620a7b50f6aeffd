"""Secondary measurements printed on bench.py's JSON line beside the headline (BASELINE.json `metric`, second half:
"encoder+NMS ms/img vs CPU ref"; configs[2] training-step kernels; configs[3] data-parallel training step).

Every function returns a JSON-able dict and never raises (an exception becomes {"error": ...}) so the headline
number cannot be lost to a secondary leg.  The CPU legs call `oracle/np_oracle.py` (the checker) as the reference's
CPU path -- measured beside the product, never used by it.
"""
import os
import time

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


def _events_ms(fn, reps, stream=None):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def _guard(fn):
    def run(*a, **k):
        try:
            return fn(*a, **k)
        except Exception as e:                       # noqa: BLE001 -- a secondary leg must not sink the headline
            return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    return run


def _encoder_pair(cfg, **over):
    from oracle import np_oracle as orc
    from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder
    kw = dict(cfg)
    kw.update(over)
    return SSDInputEncoder(**kw), orc.EncoderOracle(**kw)


@_guard
def encoder_leg(dev, B, with_cpu=True):
    """SSDInputEncoder.__call__ on B images (BASELINE configs[2]: SSD300, 21 classes, multi matching, pos 0.5 / neg 0.5,
    1-8 ground truth boxes per image, seed 7): HIP kernels (E1 iou, E2 bipartite, E3 finalize) vs the NumPy port."""
    from ssd_keras_amd import synthetic as syn
    cfg = syn.SSD300_VOC
    enc, ora = _encoder_pair(cfg, matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5)
    gt = syn.make_ground_truth(B, cfg["n_classes"], 300, 300, max_boxes=8, seed=7)
    with torch.cuda.device(dev):
        enc.encode_to_device(gt, device=dev)                                  # warm-up (uploads the anchors once)
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 20
        for _ in range(reps):
            y32, _, _ = enc.encode_to_device(gt, device=dev)
        torch.cuda.synchronize()
        wall_ms = 1e3 * (time.perf_counter() - t) / reps                      # host CSR packing + H2D of the labels + kernels
        ev_ms = _events_ms(lambda: enc.encode_to_device(gt, device=dev), reps)
        # the kernels alone: the same labels already on the device in CSR form (encode_packed: no host packing, no upload)
        gt_np, off_np, max_g = enc._pack_ground_truth(gt)
        gt_d, off_d = torch.from_numpy(gt_np).to(dev), torch.from_numpy(off_np).to(dev)
        enc.encode_packed(gt_d, off_d, gt_np.shape[0], max_g, B)
        torch.cuda.synchronize()
        kern_ms = _events_ms(lambda: enc.encode_packed(gt_d, off_d, gt_np.shape[0], max_g, B), 50)
    N, L = y32.shape[1], y32.shape[2]
    algo = B * N * L * 4 + sum(g.shape[0] for g in gt) * 5 * 8                # f32 targets written + labels read
    out = {"workload": "SSD300/VOC targets, batch %d, 1-8 GT boxes/img, multi matching, f32 output" % B,
           "gpu_ms_per_batch_wall": round(wall_ms, 4), "gpu_ms_per_batch_stream": round(ev_ms, 4),
           "gpu_ms_per_batch_kernels": round(kern_ms, 4), "gpu_ms_per_img": round(wall_ms / B, 5),
           "note": "wall / stream include the host side of SSDInputEncoder (CSR packing of the label list, checks, one upload), which "
                   "bounds a call at ~0.09 ms; `kernels` = encode_packed on device-resident labels (match_kernel + finalize_kernel)",
           "roofline": {"kernel": "match_kernel + finalize_kernel<float>", "bound": "hbm",
                        "algorithmic_bytes": algo, "achieved": round(algo / (kern_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
    if with_cpu:
        with np.errstate(invalid="ignore", divide="ignore"):
            ora(gt[:2])
            t = time.perf_counter()
            ref = ora(gt)
            cpu_ms = 1e3 * (time.perf_counter() - t)
        out["cpu"] = {"ms_per_img": round(cpu_ms / B, 4), "cores": 1, "kind": "port",
                      "sample": "oracle EncoderOracle.__call__ (NumPy port of ssd_input_encoder.py:277-418) on the same %d label "
                                "arrays, one pass" % B,
                      "speedup": round(cpu_ms / wall_ms, 1)}
        try:                                                                  # parity on the same labels (a reported flag)
            got = y32.float().cpu().numpy()
            C = got.shape[2] - 12
            out["cpu"]["hip_matches_port"] = bool(np.array_equal(got[:, :, :C], ref[:, :, :C].astype(np.float32))
                                                  and np.allclose(got, ref.astype(np.float32), rtol=1e-6, atol=1e-7))
        except Exception as exc:                                              # noqa: BLE001
            out["cpu"]["hip_matches_port"] = "error: %s: %s" % (type(exc).__name__, exc)
    return out


@_guard
def loss_leg(dev, B, with_cpu=True):
    """SSDLoss.compute_loss forward + backward (BASELINE configs[2]): y_true = the encoder's targets, y_pred = the
    trained-model-like sparse prediction tensor."""
    from oracle import np_oracle as orc
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss
    cfg = syn.SSD300_VOC
    enc, ora = _encoder_pair(cfg, matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5)
    gt = syn.make_ground_truth(B, cfg["n_classes"], 300, 300, max_boxes=8, seed=7)
    with torch.cuda.device(dev):
        y_true, _, _ = enc.encode_to_device(gt, device=dev)
        av = ora.generate_encoding_template(1)[0, :, -8:]
        y_host = syn.make_y_pred(av, B, enc.n_classes, bias=7.0, seed=1234)
        y_pred = torch.from_numpy(y_host).to(dev).requires_grad_(True)
        lf = SSDLoss(neg_pos_ratio=3, n_neg_min=0, alpha=1.0)
        loss = lf.compute_loss(y_true, y_pred)
        loss.sum().backward()
        torch.cuda.synchronize()
        with torch.no_grad():
            fwd_ms = _events_ms(lambda: lf.compute_loss(y_true, y_pred.detach()), 30)

        def fb():
            y_pred.grad = None
            lf.compute_loss(y_true, y_pred).sum().backward()
        both_ms = _events_ms(fb, 30)
    N, L = y_pred.shape[1], y_pred.shape[2]
    fwd_bytes, bwd_bytes = 2 * B * N * L * 4, 3 * B * N * L * 4
    out = {"workload": "SSDLoss(3, 0, 1.0) on (%d, %d, %d) f32" % (B, N, L),
           "fwd_ms": round(fwd_ms, 4), "fwd_bwd_ms": round(both_ms, 4),
           "roofline_fwd": {"bound": "hbm", "algorithmic_bytes": fwd_bytes, "achieved": round(fwd_bytes / (fwd_ms * 1e-3) / 1e9, 2),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fwd_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
           "roofline_fwd_bwd": {"bound": "hbm", "algorithmic_bytes": fwd_bytes + bwd_bytes,
                                "achieved": round((fwd_bytes + bwd_bytes) / (both_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round((fwd_bytes + bwd_bytes) / (both_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                "note": "includes the autograd glue (sum, grad buffers) around the two kernels"}}
    if with_cpu:
        yt = y_true.cpu().numpy()
        t = time.perf_counter()
        ref_loss = orc.ssd_loss(yt, y_host)
        ref_grad = orc.ssd_loss_grad(yt, y_host, np.ones((B,), dtype=np.float32))
        cpu_ms = 1e3 * (time.perf_counter() - t)
        out["cpu"] = {"fwd_bwd_ms": round(cpu_ms, 2), "cores": 1, "kind": "port",
                      "sample": "oracle ssd_loss + ssd_loss_grad (NumPy restatement of keras_ssd_loss.py:98-211; TensorFlow absent) "
                                "on the same tensors, one pass", "speedup": round(cpu_ms / both_ms, 1)}
        try:                                                                  # north_star tolerance 1e-4 (a reported flag)
            with torch.cuda.device(dev):
                y_pred.grad = None
                got_loss = lf.compute_loss(y_true, y_pred)
                got_loss.sum().backward()
            out["cpu"]["hip_loss_within_1e-4_of_port"] = bool(np.allclose(got_loss.detach().cpu().numpy(), ref_loss, rtol=1e-4, atol=1e-6))
            # a hard negative sitting exactly on the k-th loss value may be kept by one side only (device logf vs NumPy log):
            # its row of the gradient then differs; everything else is within the tolerance
            out["cpu"]["hip_grad_elements_outside_1e-4"] = int((~np.isclose(y_pred.grad.cpu().numpy(), ref_grad, rtol=1e-4, atol=1e-6)).sum())
        except Exception as exc:                                              # noqa: BLE001
            out["cpu"]["hip_loss_within_1e-4_of_port"] = "error: %s: %s" % (type(exc).__name__, exc)
    return out


@_guard
def sparse_decode_leg(dev, B, with_cpu=True):
    """decode_detections on the trained-model-like SPARSE tensor (background logit +7, ~6.5k candidates/img at 0.01;
    SURVEY 8d config 2) -- the regime a trained detector is in; the headline step decodes random-init (dense) output."""
    from oracle import np_oracle as orc
    from ssd_keras_amd import _native as nat
    from ssd_keras_amd import synthetic as syn
    cfg = syn.SSD300_VOC
    _, ora = _encoder_pair(cfg)
    av = ora.generate_encoding_template(1)[0, :, -8:]
    y_host = syn.make_y_pred(av, B, ora.n_classes, bias=7.0, seed=1234)
    with torch.cuda.device(dev):
        y = torch.from_numpy(y_host).to(dev)
        N, C = y.shape[1], y.shape[2] - 12
        dkw = dict(conf_thresh=0.01, iou_thresh=0.45, top_k=200, nms_cap=0, class_agnostic=False, semantics=nat.SEM_NUMPY,
                   coords="centroids", normalize_coords=True, img_height=300, img_width=300, border_pixels="half",
                   out_dtype=nat.F64, out_rows=200)
        outs = nat.decode(y, **dkw)
        torch.cuda.synchronize()
        stage = {}
        for name, mask in (("scan_kernel", 1), ("nms_kernel", 2), ("topk_kernel", 4), ("decode_path", 7)):
            nat.decode(y, stages=mask, outputs=outs, **dkw)
            torch.cuda.synchronize()
            stage[name] = _events_ms(lambda m=mask: nat.decode(y, stages=m, outputs=outs, **dkw), 50)
    algo = B * (N * (C + 12) * 4 + 200 * 6 * 8)
    out = {"workload": "decode_detections (NumPy semantics, f64 rows) on the sparse SSD300 tensor, batch %d, conf 0.01 / "
                       "NMS 0.45 / top-200" % B,
           "kernel_ms": {k: round(v, 5) for k, v in stage.items()}, "gpu_ms_per_img": round(stage["decode_path"] / B, 5),
           "roofline": {"bound": "hbm", "algorithmic_bytes": algo, "achieved": round(algo / (stage["decode_path"] * 1e-3) / 1e9, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / (stage["decode_path"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "scan_kernel_GBps": round(B * N * (C + 12) * 4 / (stage["scan_kernel"] * 1e-3) / 1e9, 1)}}
    if with_cpu:
        kw = dict(confidence_thresh=0.01, iou_threshold=0.45, top_k=200, normalize_coords=True, img_height=300, img_width=300)
        t = time.perf_counter()
        orc.decode_detections(y_host[:2], **kw)
        per = (time.perf_counter() - t) / 2
        n_img = int(max(2, min(B, round(10.0 / max(per, 1e-3)))))
        t = time.perf_counter()
        ref = orc.decode_detections(y_host[:n_img], exp_mode="det", **kw)     # "det": the exp shared with the kernel
        cpu_ms = 1e3 * (time.perf_counter() - t) / n_img
        out["cpu"] = {"ms_per_img": round(cpu_ms, 3), "cores": 1, "kind": "port",
                      "sample": "oracle decode_detections on the first %d of %d images" % (n_img, B),
                      "speedup": round(cpu_ms / (stage["decode_path"] / B), 1)}
        try:                                                                  # parity on the sample (a reported flag)
            from oracle import parity as par
            rows, count = outs[0].cpu().numpy(), outs[1].cpu().numpy()
            ref_all = orc.decode_detections(y_host[:n_img], exp_mode="det", **dict(kw, top_k="all"))
            out["cpu"]["hip_vs_port_on_the_sample"] = par.decode_parity([rows[b, :int(count[b])] for b in range(n_img)], ref_all, 200)
        except Exception as exc:                                              # noqa: BLE001
            out["cpu"]["hip_vs_port_on_the_sample"] = "error: %s: %s" % (type(exc).__name__, exc)
    return out


@_guard
def ssd512_decode_leg(dev, with_cpu=True):
    """BASELINE configs[4] / SURVEY 8d config 5: SSD512, 81 classes (COCO shape), batch 16, 24564 anchors -> (16, 24564, 93).
    Sparse (background logit +7, ~45 k candidates/img at 0.01) timed beside the NumPy port on a bounded sample; dense (bias 0,
    conf 0.001: up to 1.97 M candidates/img) GPU-only with size-independent property checks."""
    from oracle import np_oracle as orc
    from oracle import parity as par
    from ssd_keras_amd import _native as nat
    from ssd_keras_amd import synthetic as syn
    cfg = syn.SSD512_COCO
    _, ora = _encoder_pair(cfg)
    av = ora.generate_encoding_template(1)[0, :, -8:]
    B = 16
    out = {"workload": "SSD512, 81 classes, batch 16, 24564 anchors: decode_detections (NumPy semantics, f64 rows), NMS 0.45 / top-200"}
    with torch.cuda.device(dev):
        for name, bias, thr in (("sparse_bias7_conf0.01", 7.0, 0.01), ("dense_bias0_conf0.001", 0.0, 0.001)):
            y_host = syn.make_y_pred(av, B, ora.n_classes, bias=bias, seed=1234)
            y = torch.from_numpy(y_host).to(dev)
            N, C = y.shape[1], y.shape[2] - 12
            dkw = dict(conf_thresh=thr, iou_thresh=0.45, top_k=200, nms_cap=0, class_agnostic=False, semantics=nat.SEM_NUMPY,
                       coords="centroids", normalize_coords=True, img_height=512, img_width=512, border_pixels="half",
                       out_dtype=nat.F64, out_rows=200)
            outs = nat.decode(y, **dkw)
            torch.cuda.synchronize()
            stage = {}
            for kn, mask in (("scan_kernel", 1), ("nms_kernel", 2), ("topk_kernel", 4), ("decode_path", 7)):
                nat.decode(y, stages=mask, outputs=outs, **dkw)
                torch.cuda.synchronize()
                stage[kn] = _events_ms(lambda m=mask: nat.decode(y, stages=m, outputs=outs, **dkw), 20)
            algo = B * (N * (C + 12) * 4 + 200 * 6 * 8)
            dom = max(("scan_kernel", "nms_kernel", "topk_kernel"), key=lambda k: stage[k])
            leg = {"candidates_per_img": int((y_host[:, :, 1:C] > thr).sum() // B),
                   "kernel_ms": {k: round(v, 5) for k, v in stage.items()}, "gpu_ms_per_img": round(stage["decode_path"] / B, 5),
                   "roofline": {"kernel": dom, "bound": "hbm", "algorithmic_bytes_per_launch": algo,
                                "achieved": round(algo / (stage[dom] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(algo / (stage[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                "decode_path_GBps": round(algo / (stage["decode_path"] * 1e-3) / 1e9, 2),
                                "scan_kernel_GBps": round(B * N * (C + 12) * 4 / (stage["scan_kernel"] * 1e-3) / 1e9, 1)}}
            rows, count = outs[0].cpu().numpy(), outs[1].cpu().numpy()
            # size-independent properties: every row is a real (class, anchor) candidate above the threshold, rows of one class
            # do not overlap by more than the NMS threshold, the decode is idempotent
            outs2 = nat.decode(y, **dkw)
            props = bool(torch.equal(outs2[0], outs[0]) and torch.equal(outs2[1], outs[1]))
            for b in range(0, B, 5):
                r = rows[b, :int(count[b])]
                cls = r[:, 0].astype(int)
                props = props and bool(np.all((cls >= 1) & (cls < C)) and np.all(r[:, 1] > thr))
                for cl in np.unique(cls)[:6]:
                    rc = r[cls == cl]
                    props = props and bool(np.isin(rc[:, 1].astype(np.float32), y_host[b, :, cl]).all())
                    if rc.shape[0] > 1:
                        iou = orc.iou(rc[:, 2:], rc[:, 2:], coords="corners", mode="outer_product")
                        np.fill_diagonal(iou, 0.0)
                        props = props and bool(iou.max() <= 0.45)
            leg["properties_hold"] = props
            if with_cpu and bias > 0:
                kw = dict(confidence_thresh=thr, iou_threshold=0.45, top_k=200, normalize_coords=True, img_height=512, img_width=512)
                t = time.perf_counter()
                orc.decode_detections(y_host[:1], **kw)
                per = time.perf_counter() - t
                n_img = int(max(2, min(B, round(12.0 / max(per, 1e-3)))))
                t = time.perf_counter()
                ref_all = orc.decode_detections(y_host[:n_img], exp_mode="det", **dict(kw, top_k="all"))
                cpu_ms = 1e3 * (time.perf_counter() - t) / n_img
                leg["cpu"] = {"ms_per_img": round(cpu_ms, 3), "cores": 1, "kind": "port",
                              "sample": "oracle decode_detections on the first %d of %d images" % (n_img, B),
                              "speedup": round(cpu_ms / (stage["decode_path"] / B), 1),
                              "hip_vs_port_on_the_sample": par.decode_parity([rows[b, :int(count[b])] for b in range(n_img)], ref_all, 200)}
            out[name] = leg
            del y, outs, outs2
    return out


@_guard
def fp32_forward_leg(dev, B, reps=3):
    """The reference's convolutions are float32: the same SSD300 forward at reference precision (MIOpen fp32; none of the bf16
    MFMA kernels engage), as the companion of the bf16 headline.  62.747 GFLOP/img against the 157.3 TFLOP/s fp32 matrix peak."""
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.models.keras_ssd300 import ssd_300
    cfg = syn.SSD300_VOC
    torch.manual_seed(1234)
    model = ssd_300((300, 300, 3), cfg["n_classes"], mode="inference", scales=cfg["scales"],
                    aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"],
                    confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400).to(dev)
    model = model.to(memory_format=torch.channels_last).eval()
    images = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).to(dev)
    with torch.cuda.device(dev), torch.no_grad():
        for _ in range(2):
            model.raw_predictions(images)
        torch.cuda.synchronize()
        fwd_ms = _events_ms(lambda: model.raw_predictions(images), reps)
        step_ms = _events_ms(lambda: model(images), reps)
    tf = B * 62.747 / 1e3 / (fwd_ms * 1e-3)
    return {"bound": "mfma", "dtype": "fp32", "forward_ms": round(fwd_ms, 3), "step_ms_fwd_plus_decode": round(step_ms, 3),
            "images_per_sec": round(B / (step_ms * 1e-3), 1), "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s",
            "frac": round(tf / 157.3, 4), "note": "MIOpen float32 convolutions (PyTorch-ROCm), HIP decode; not the timed headline"}


@_guard
def fp32x3_forward_leg(dev, B, fp32_leg=None, reps=3):
    """The float32 model (the reference's precision) through the float16 x 3 MFMA convolutions (models/precise.py,
    ssdhip_conv2d_x3_nhwc_f16) + the HIP DecodeDetections: the reference-precision companion of the bf16 headline, beside MIOpen's
    float32 convolutions (`conv_roofline_fp32`).  Parity flags: predictions against the float32 framework forward of the same model."""
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.models.keras_ssd300 import ssd_300
    from ssd_keras_amd.models.precise import PreciseForward
    cfg = syn.SSD300_VOC
    torch.manual_seed(1234)
    model = ssd_300((300, 300, 3), cfg["n_classes"], mode="inference", scales=cfg["scales"],
                    aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"],
                    confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400).to(dev)
    model = model.to(memory_format=torch.channels_last).eval()
    with torch.no_grad():                                                    # the tamed heads of tests/test_precise_gpu.py: a raw
        for head in model.conf_heads:                                        # He-init softmax is saturated (logits ~1e4), and two float32
            head.weight.mul_(1e-3)                                           # forwards of it (CPU vs MIOpen) already differ by 4e-4
            head.bias.view(-1, cfg["n_classes"] + 1)[:, 0] = 4.0
        for head in model.loc_heads:
            head.weight.mul_(1e-3)
    images = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).to(dev)
    pf = PreciseForward(model)
    launch, graph_info = "eager", {}
    with torch.cuda.device(dev), torch.no_grad():
        pred = pf(images)
        torch.cuda.synchronize()
        fwd_ms = _events_ms(lambda: pf(images), reps)
        step_ms = _events_ms(lambda: pf(images, decode=True), reps)           # DecodeDetections straight from the float32 head maps
        ref = model.raw_predictions(images).float()                          # MIOpen float32 forward of the same model
        # the step as ONE HIP graph (round 5): model.precise() makes this path the model's forward, model.graphed() captures it
        try:
            model.__dict__["_precise"] = pf
            runner = model.graphed(images)
            out_g = runner(images).clone()
            out_e = pf(images, decode=True)
            for _ in range(8):                                                # untimed: the clock ramp behind the capture (bench.py --graph-warmup)
                runner(images)
            graph_ms = _events_ms(lambda: runner(images), max(reps, 10))
            graph_info = {"graph_step_ms": round(graph_ms, 3), "graph_output_equals_eager": bool(torch.equal(out_g, out_e)),
                          "eager_step_ms": round(step_ms, 3)}
            if graph_info["graph_output_equals_eager"] and graph_ms < step_ms:
                step_ms, launch = graph_ms, "hip_graph (forward + DecodeDetections captured once; output == the eager step's)"
        except Exception as exc:                                              # noqa: BLE001 -- the eager number stands
            launch = "eager (graph capture failed: %s)" % (repr(exc)[:120])
        finally:
            model.__dict__["_precise"] = None
    C = pred.shape[2] - 12
    finite = torch.isfinite(ref[:, :, :C + 4]) & torch.isfinite(pred[:, :, :C + 4])
    d = (pred[:, :, :C + 4] - ref[:, :, :C + 4]).abs()
    rel = d / (ref[:, :, :C + 4].abs() + 1.0)
    tf = 3 * B * 62.747 / 1e3 / (fwd_ms * 1e-3)
    out = {"bound": "mfma", "dtype": "float16 hi/lo pairs, hi.hi + hi.lo + lo.hi, float32 accumulation (float32-grade: 2^-22 per product)",
           "forward_ms": round(fwd_ms, 3), "step_ms_fwd_plus_decode": round(step_ms, 3), "images_per_sec": round(B / (step_ms * 1e-3), 1),
           "launch": launch, "graph": graph_info,
           "achieved": round(tf, 2), "peak": 2500.0, "unit": "TFLOP/s of float16 MFMA work (3 x 62.747 GFLOP/img)", "frac": round(tf / 2500.0, 4),
           "vs_framework_float32": {"max_abs_diff_class_probabilities": float(d[:, :, :C][finite[:, :, :C]].max().item()),
                                    "max_rel_diff_offsets": float(rel[:, :, C:][finite[:, :, C:]].max().item()),
                                    "non_finite_in_either": int((~finite).sum().item())},
           "note": "3x3 'same' convolutions with 128-multiple channels on the slab kernel (csrc/ssdhip_convh.hip, X3); round 6: small maps "
                   "(conv5_x, fc6, fc7, conv6_x) on the image-resident kernel's X3 form (csrc/ssdhip_convimg.hip), pool4 / pool5 on pair "
                   "maps (ssdhip_x3_maxpool_nhwc); the rest on the implicit-GEMM kernel (csrc/ssdhip_conv.hip, X3), conv1_1 (K = 27) float32 "
                   "vector arithmetic (ssdhip_conv1_1_x3_nhwc), L2Normalization in float32; Reshape / Concatenate / softmax / AnchorBoxes + "
                   "DecodeDetections in one libssdhip pipeline straight from the float32 head maps; parity flags on tamed heads "
                   "(filters x 1e-3, background bias 4: an unsaturated softmax)"}
    if isinstance(fp32_leg, dict) and fp32_leg.get("images_per_sec"):
        out["speedup_over_miopen_float32"] = round(out["images_per_sec"] / fp32_leg["images_per_sec"], 3)
    return out


@_guard
def other_models_forward_leg(dev, reps=5):
    """SSD512 and SSD7 forward + DecodeDetections (mode='inference') at the README's batch 8 and at batch 32, beside the reference's
    published whole-model figures (README.md:107-124: SSD512 25 FPS, SSD7 216 FPS at batch 8 on a GTX 1070 mobile -- other hardware,
    Pascal VOC weights; here random-init weights on synthetic images, so the NMS works on the dense regime).  bf16 backbone, the
    model's own fused path (models/_common.py); eager launches, events on the stream.  VERDICT r4 item 7."""
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.models.keras_ssd512 import ssd_512
    from ssd_keras_amd.models.keras_ssd7 import build_model
    out = {}

    def run(name, model, size, batches, ref_fps):
        model = model.to(dev).to(memory_format=torch.channels_last).eval().to(torch.bfloat16)
        res = {"reference_fps_batch8_gtx1070_mobile": ref_fps}
        for b in batches:
            images = torch.from_numpy(np.random.RandomState(b).randint(0, 256, size=(b, size, size, 3)).astype(np.float32)).to(dev)
            with torch.cuda.device(dev), torch.no_grad():
                for _ in range(3):
                    y = model(images)
                torch.cuda.synchronize()
                ms = _events_ms(lambda: model(images), reps)
            res["batch%d" % b] = {"ms_per_step": round(ms, 4), "images_per_sec": round(b / (ms * 1e-3), 1), "output": list(y.shape)}
        out[name] = res

    c5 = syn.SSD512_COCO
    torch.manual_seed(5)
    run("ssd512_voc_21_classes", ssd_512((512, 512, 3), 20, mode="inference", scales=[0.07, 0.15, 0.3, 0.45, 0.6, 0.75, 0.9, 1.05],
                                         aspect_ratios_per_layer=c5["aspect_ratios_per_layer"], steps=c5["steps"], offsets=c5["offsets"],
                                         confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400), 512, (8, 32), 25)
    c7 = syn.SSD7_300
    torch.manual_seed(7)
    run("ssd7_300x300_5_classes", build_model((300, 300, 3), c7["n_classes"], mode="inference", scales=c7["scales"],
                                              aspect_ratios_global=c7["aspect_ratios_global"], variances=c7["variances"],
                                              normalize_coords=True, subtract_mean=127.5, divide_by_stddev=127.5,
                                              confidence_thresh=0.5, iou_threshold=0.45, top_k=200, nms_max_output_size=400), 300, (8, 32), 216)
    return out


@_guard
def train_leg(dev, rank, world, B, steps=6, warmup=3, tame=True):
    """BASELINE configs[2]/[3]: one SSD300 training step per rank = SSDInputEncoder (HIP) -> forward (bf16 autocast,
    fp32 master weights) -> SSDLoss (HIP, local hard-negative mining) -> backward -> RCCL gradient all-reduce (DDP,
    25 MB buckets overlapped with backward) -> SGD(momentum 0.9).  Weak scaling: B images per rank.
    `tame`: predictor-head filters x 1e-2 and background bias + 4, so that the softmax of the random-init model is neither saturated
    nor uniform and the timed steps are a descending optimisation (raw He-init on 0..255 inputs clips most losses at -log(1e-15):
    a tie-saturated hard-negative select and a loss that grows at any usable learning rate -- reported as `raw_init` at N = 1).
    Learning rate 1e-7: the first gradient has norm 4e4 (un-normalised 0..255 inputs through a He-init VGG); 24 eager steps go
    34.4 -> 30.1 at 1e-7, 34.4 -> 745 -> 113 at 1e-5 (chaotic: one graphed run ended non-finite) -- tools/debug_train.py, r03zc."""
    from ssd_keras_amd import distributed as dp
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss
    from ssd_keras_amd.models.keras_ssd300 import ssd_300
    from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder
    cfg = syn.SSD300_VOC
    torch.manual_seed(4321)                                                   # same initial weights on every rank
    model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", l2_regularization=0.0005, scales=cfg["scales"],
                    aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"]).to(dev)
    model = model.to(memory_format=torch.channels_last).train()
    if tame:
        with torch.no_grad():
            for head in model.conf_heads:
                head.weight.mul_(1e-2)
                head.bias.view(-1, cfg["n_classes"] + 1)[:, 0] = 4.0
            for head in model.loc_heads:
                head.weight.mul_(1e-2)
    n_params = sum(p.numel() for p in model.parameters())
    ddp = dp.data_parallel(model, dev)
    # Keras adds l2(5e-4) * sum(W^2) over the conv kernels to the loss (keras_ssd300.py:274): gradient 2 * l2 * W == SGD weight
    # decay 1e-3 on the kernels only
    decay = [p for p in model.parameters() if p.dim() > 1]
    plain = [p for p in model.parameters() if p.dim() <= 1]
    # keras.optimizers.SGD(lr, momentum=0.9) of ssd300_training.ipynb:169 as ONE launch over all parameters (ssd_keras_amd/optimizers.py;
    # SSD_TRAIN_TORCH_SGD=1: the framework's optimizer, for A/B runs)
    from ssd_keras_amd.optimizers import SGD as FusedSGD
    opt_cls = torch.optim.SGD if os.environ.get("SSD_TRAIN_TORCH_SGD", "0") == "1" else FusedSGD
    opt = opt_cls([{"params": decay, "weight_decay": 1e-3}, {"params": plain, "weight_decay": 0.0}], lr=1e-7, momentum=0.9)
    enc = SSDInputEncoder(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
    gt = syn.make_ground_truth(B, cfg["n_classes"], 300, 300, max_boxes=8, seed=7 + rank)
    images = torch.from_numpy(np.random.RandomState(100 + rank).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).to(dev)
    lf = SSDLoss(neg_pos_ratio=3, n_neg_min=0, alpha=1.0)
    last = {}
    trace = [] if os.environ.get("SSD_TRAIN_TRACE", "0") == "1" else None     # per-step losses (one host sync per step: not for timing)

    def step():
        y_true, _, _ = enc.encode_to_device(gt, device=dev)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y_pred = ddp(images)
        loss = lf.compute_loss(y_true, y_pred.float()).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        last["loss"] = loss.detach()
        last.setdefault("first", loss.detach())
        if trace is not None:
            trace.append(("eager", round(float(loss), 4)))

    # One GPU: the step is ~600 launches and the eager host loop cannot issue them as fast as the GPU retires them (measured:
    # 19.8 ms per step eager with 11 ms of kernels, profiles/r02g_train_timeline.json), so forward + loss + backward are
    # captured ONCE into a HIP graph (static buffers for the images and the encoder's targets) and replayed; the encoder (its labels
    # arrive from the host every step) and the optimizer step (see capture()) stay outside.  With DDP (N > 1) the step stays eager: RCCL's bucketed all-reduce
    # hooks inside a capture could not be exercised on this single-GPU box.
    graph = None
    how = "eager"
    # Round 6: the one-launch SGD (ssd_keras_amd/optimizers.py) is captured with the rest -- the step is ONE graph launch + the encoder
    # (SSD_TRAIN_GRAPH_OPT=0, or the framework's optimizer: the update is issued eagerly after each replay as in rounds 3-5)
    opt_in_graph = os.environ.get("SSD_TRAIN_GRAPH_OPT", "1") == "1" and opt_cls is FusedSGD

    def capture():
        nonlocal graph
        y_static, _, _ = enc.encode_to_device(gt, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        def warm_step():                                                      # (a function: its autograd graph dies with its locals)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y_pred = ddp(images)
            loss = lf.compute_loss(y_static, y_pred.float()).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()

        with torch.cuda.stream(side):                                         # warm the allocator / autotune off the default stream
            for _ in range(2):
                warm_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        # The optimizer step stays OUTSIDE the capture.  With torch.optim.SGD's multi-tensor kernels inside it, the replays that follow
        # the first device-wide synchronize apply corrupted updates: replays 1-2 follow the eager trajectory, then the loss jumps 30.5 ->
        # 96 (the same value on every box), later now and then to NaN -- the round-2 leg's "diverging" loss and r03z's NaN were this,
        # not the learning rate.  Bisected with tools/debug_graph_leg.py (profiles/r03zk_train_graph_bisect.txt): it takes the
        # libssdhip autograd functions AND the captured optimizer AND a synchronize between replays; no single kernel of ours (dgrad,
        # pooling, ReLU / bias, L2Normalization, weight shadows, the loss -- each switched off in turn), not the encoder, not a stale
        # autograd graph.  The mechanism inside the framework's graph memory handling is not established; the optimizer step issued
        # eagerly after each replay follows the eager trajectory step for step.
        import gc
        gc.collect()
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y_pred = ddp(images)
            loss_static = lf.compute_loss(y_static, y_pred.float()).mean()
            loss_static.backward()                        # writes the .grad tensors allocated here: opt.step() reads them
            if opt_in_graph:
                opt.step()
        torch.cuda.synchronize()
        graph = (g, y_static, loss_static)

    def graph_step():
        g, y_static, loss_static = graph
        y_true, _, _ = enc.encode_to_device(gt, device=dev)
        y_static.copy_(y_true)
        g.replay()
        if not opt_in_graph:
            opt.step()
        last["loss"] = loss_static.detach()
        if trace is not None:
            trace.append(("graph", round(float(loss_static), 4)))
            torch.cuda.synchronize()                     # a device-wide synchronize between replays was part of the round-3 failure

    eager_ms = None
    with torch.cuda.device(dev):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        # the eager step's own time, at every N: what a graphed N = 1 step must be compared with when scaling efficiency is read off
        # (under DDP the step is eager)
        t = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - t) / steps
        run = step
        # Graph replay of the step only when the runtime's graph packet capture is off (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the
        # environment before HIP initialises): with it on (the default of this ROCm 7.2 stack) hipMemsetAsync nodes inside a replayed graph
        # do not take effect on the replays that follow a device-wide synchronize -- libssdhip no longer issues any (csrc/ssdhip_math.h
        # zero_async, profiles/r04l_loss_graph_memset_node.txt), but the framework's remaining backward kernels (MIOpen weight gradients
        # of the predictor heads / extra layers) still do, and their gradients go to inf on the fourth replay
        # (profiles/r04m_graph_rounds_after_zero_kernel.txt).  The step is GPU-bound: eager costs 0.5 %.
        # Round 6: the backward pass holds no framework convolution any more (models/_common.py, _conv_input_weight_grads: the dilated /
        # strided / 'valid' 3 x 3 layers have their own kernels), hence no memset node: the whole step replays under the runtime's
        # defaults (tests/test_train_graph_gpu.py); with SSDHIP_NO_TAPS_BWD=1 or SSDHIP_NO_OWN_WGRAD=1 (MIOpen back in) only with packet capture off.
        miopen_back = any(os.environ.get(k, "0") == "1" for k in ("SSDHIP_NO_TAPS_BWD", "SSDHIP_NO_OWN_WGRAD", "SSDHIP_NO_OWN_DGRAD"))
        safe_graph = os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "") == "0" or not miopen_back
        if world == 1 and os.environ.get("SSD_TRAIN_GRAPH", "1" if safe_graph else "0") == "1":
            try:
                capture()
                for _ in range(int(os.environ.get("SSD_TRAIN_GRAPH_WARMUP", "6"))):      # untimed replays: the clock ramp behind the capture (2 in rounds 3-6)
                    graph_step()
                torch.cuda.synchronize()
                run, how = graph_step, ("ONE hipGraph replay per step (forward + SSDLoss + backward + SGD captured once); the encoder eager"
                                        if opt_in_graph else "hipGraph replay (forward + loss + backward captured once) + eager SGD step")
            except Exception as exc:                                          # noqa: BLE001 -- fall back to the eager step
                how = "eager (graph capture failed: %s: %s)" % (type(exc).__name__, str(exc)[:120])
                torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        prof = None
        if os.environ.get("SSD_TRAIN_HOST_PROFILE"):                      # tools aid: where the host spends the step (cProfile of the timed loop)
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t = time.perf_counter()
        for _ in range(steps):
            run()
        issued = time.perf_counter() - t                                  # the host has issued every launch; the GPU may still be running
        torch.cuda.synchronize()
        if prof is not None:
            import io
            import pstats
            prof.disable()
            buf = io.StringIO()
            pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(45)
            with open(os.environ["SSD_TRAIN_HOST_PROFILE"], "w") as fh:
                fh.write("host issue time per step %.3f ms (of %.3f ms per step)\n" % (1e3 * issued / steps, 1e3 * (time.perf_counter() - t) / steps))
                fh.write(buf.getvalue())
        if world > 1:
            torch.distributed.barrier()
        mine = time.perf_counter() - t
        elapsed = dp.max_over_ranks(mine, device=dev)
        fastest = -dp.max_over_ranks(-mine, device=dev)
        eager_max = dp.max_over_ranks(eager_ms, device=dev)
    n_buckets = 0
    if world > 1:
        try:
            n_buckets = len(ddp.reducer._get_zeros_like_grad_buckets())          # private, informational only
        except Exception:                                                     # noqa: BLE001
            n_buckets = -(-4 * n_params // (25 * 1024 * 1024))
    out_raw = None
    if world == 1 and tame and os.environ.get("SSD_TRAIN_RAW", "1") == "1":
        del ddp, opt, model
        torch.cuda.empty_cache()
        raw = train_leg(dev, rank, world, B, steps=steps, warmup=warmup, tame=False)
        if isinstance(raw, dict):
            out_raw = {k: raw.get(k) for k in ("images_per_sec", "ms_per_step", "eager_ms_per_step", "first_loss", "final_loss", "launch", "error")
                       if raw.get(k) is not None}
    return {"regime": "tamed heads (filters x 1e-2, background bias + 4), lr 1e-7" if tame else "raw He-normal init, lr 1e-7",
            "raw_init": out_raw, "loss_trace": trace, "eager_ms_per_step": round(eager_max, 3),
            "rank_step_ms_min_max": [round(1e3 * fastest / steps, 3), round(1e3 * elapsed / steps, 3)],
            "allreduce_buckets": n_buckets, "bucket_cap_mb": 25,
            "workload": "SSD300 VGG-16 training step, 21 classes, batch %d per GPU (global %d), bf16 autocast + fp32 master "
                        "weights, SGD momentum 0.9; HIP encoder + HIP SSDLoss; %s" % (
                            B, B * world, "DDP/RCCL gradient all-reduce (25 MB buckets)" if world > 1 else "single GPU, no collective"),
            "images_per_sec": round(world * B * steps / elapsed, 2), "ms_per_step": round(1e3 * elapsed / steps, 3),
            "steps": steps, "warmup": warmup, "n_gpus": world, "scaling": "weak", "parameters": n_params, "launch": how,
            "allreduce_bytes_per_step": 4 * n_params if world > 1 else 0, "first_loss": float(last["first"].item()), "final_loss": float(last["loss"].item()),
            "note": "random He-normal init on 0..255 inputs (no pretrained VGG here); `eager_ms_per_step` is the same step issued "
                    "launch by launch (what N > 1 runs under DDP) -- compare like with like when reading scaling efficiency"}


@_guard
def augmentation_leg(dev, B, with_cpu=True):
    """SURVEY 8f row 4, image half: the photometric distortions of the original-SSD chain (one launch per batch) and the resize to the
    network input on device-resident VOC-sized uint8 batches, against the NumPy restatement (oracle/np_image.py: the reference's own
    expressions + OpenCV's published conversions) on a bounded sample.  Python side included in the GPU times."""
    from ssd_keras_amd.data_generator import _image_ops as iop
    from ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd import SSDPhotometricDistortions
    rng = np.random.RandomState(0)
    host = rng.randint(0, 256, size=(B, 375, 500, 3)).astype(np.uint8)
    batch = torch.from_numpy(host).to(dev)
    d = SSDPhotometricDistortions()
    state = np.random.get_state()
    np.random.seed(0)
    progs = [d.draw() for _ in range(B)]
    np.random.set_state(state)
    with torch.cuda.device(dev):
        out = iop.run_batch(batch, progs)
        t_pho = _events_ms(lambda: iop.run_batch(batch, progs), 20)
        res = {}
        for name, interp in (("nearest", 0), ("linear", 1), ("cubic", 2), ("area", 3), ("lanczos4", 4)):
            iop.resize(batch, 300, 300, interp)
            res[name] = round(_events_ms(lambda: iop.resize(batch, 300, 300, interp), 10), 4)
        small = iop.resize(batch, 300, 300, 1)
    leg = {"workload": "batch %d of 375 x 500 x 3 uint8 images resident on the device" % B,
           "photometric_ms_per_batch": round(t_pho, 4), "photometric_images_per_sec": round(B / (t_pho * 1e-3)),
           "photometric_GBps_read_plus_write": round(2 * batch.numel() / (t_pho * 1e-3) / 1e9, 1),
           "resize_to_300x300_ms_per_batch": res}
    # the whole chain as a device pipeline (SSDDataAugmentation.augment_batch): wall clock including the host's random draws, label
    # arithmetic, patch validation round trips and tap tables -- the number an input pipeline sees
    try:
        from ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd import SSDDataAugmentation
        aug = SSDDataAugmentation(img_height=300, img_width=300)
        labels = []
        for _ in range(B):
            n = rng.randint(1, 6)
            x0, y0 = rng.randint(0, 400, size=n), rng.randint(0, 280, size=n)
            labels.append(np.stack([rng.randint(1, 21, size=n), x0, y0, x0 + rng.randint(20, 100, size=n), y0 + rng.randint(20, 90, size=n)], axis=1))
        state = np.random.get_state()
        np.random.seed(1)
        with torch.cuda.device(dev):
            def timed(fn, reps):
                fn()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / reps
            # round 5: one seed per image -- the host makes the photometric draws, ONE launch (a wave per image on that image's NumPy
            # MT19937 stream) takes every other decision of the chain, a second builds the tap tables, the gather launch does the pixels
            dt = timed(lambda: aug.augment_batch(batch, labels, seeds=np.random.randint(0, 2 ** 31 - 1, size=B)), 10)
            # round 6: ONE global np.random stream across the batch (the reference's generator semantics), decided on the device by a
            # single wave walking the images in order, photometric draws included (ssdhip_ssd_augment_decide_stream)
            dt_stream = timed(lambda: aug.augment_batch(batch, labels), 10)
            # parity of that path on this batch: == the per-image chain on the same stream (first 4 images + the stream's end)
            np.random.seed(123)
            s_img, s_lab = aug.augment_batch(batch, labels)
            s_next = np.random.uniform()
            s_img = s_img.cpu().numpy()
            np.random.seed(123)
            same_stream = True
            for i in range(B):
                wi, wl = aug(host[i], labels[i])
                if i < 4:
                    same_stream = same_stream and np.array_equal(wi, s_img[i]) and np.array_equal(wl, s_lab[i])
            same_stream = same_stream and np.random.uniform() == s_next
            # the round-4 form of the same semantics: the chain's decisions per image on the host (a GPU round trip per sampling round)
            os.environ["SSDHIP_AUG_HOST_STREAM"] = "1"
            try:
                dt_host = timed(lambda: aug.augment_batch(batch, labels), 3)
            finally:
                os.environ.pop("SSDHIP_AUG_HOST_STREAM", None)
            # parity on this batch: image i == the per-image chain under np.random.seed(seed_i), bit for bit (sample of 4 images)
            seeds = np.random.randint(0, 2 ** 31 - 1, size=B)
            got_img, got_lab = aug.augment_batch(batch, labels, seeds=seeds)
            got_img = got_img.cpu().numpy()
            same = True
            for i in range(4):
                np.random.seed(int(seeds[i]))
                wi, wl = aug(host[i], labels[i])
                same = same and np.array_equal(wi, got_img[i]) and np.array_equal(wl, got_lab[i])
        np.random.set_state(state)
        leg["augment_batch_ms_per_batch_wall"] = round(1e3 * dt, 3)
        leg["augment_batch_images_per_sec"] = round(B / dt, 1)
        leg["augment_batch_equals_the_per_image_chain_under_each_seed"] = bool(same)
        leg["augment_batch_global_stream_images_per_sec"] = round(B / dt_stream, 1)
        leg["augment_batch_global_stream_ms_per_batch_wall"] = round(1e3 * dt_stream, 3)
        leg["augment_batch_global_stream_equals_the_per_image_chain"] = bool(same_stream)
        leg["augment_batch_global_stream_host_decisions_images_per_sec"] = round(B / dt_host, 1)
        leg["augment_batch_note"] = ("seeds= : the host makes each image's photometric draws; ssdhip_ssd_augment_decide (a wave per image on "
                                     "that image's NumPy MT19937 stream) takes the chain's other decisions and does the label arithmetic, "
                                     "ssdhip_augment_taps builds the tap tables, one gather launch the pixels; wall clock incl. the label "
                                     "download.  Without seeds (ONE global stream across the batch: the reference's generator loop, "
                                     "object_detection_2d_data_generator.py:1050-1089) a single wave walks the images in order on that stream "
                                     "and takes every decision incl. the photometric ones (ssdhip_ssd_augment_decide_stream), np.random "
                                     "continues behind the batch: `global_stream`; the same semantics decided on the host: `host_decisions`")
    except Exception as exc:                                                  # noqa: BLE001 -- a companion figure
        leg["augment_batch_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:160])
    if with_cpu:
        from oracle import np_image as npi
        n = 2
        t = time.perf_counter()
        want = []
        for i in range(n):
            enc = iop.encode(progs[i])
            want.append(npi.run_program(host[i], enc[0], enc[1]))
        cpu_pho = (time.perf_counter() - t) / n
        t = time.perf_counter()
        want_small = npi.resize(host[0], (300, 300), 1)
        cpu_res = time.perf_counter() - t
        leg["cpu"] = {"photometric_ms_per_img": round(1e3 * cpu_pho, 2), "resize_linear_ms_per_img": round(1e3 * cpu_res, 2), "cores": 1, "kind": "port",
                      "sample": "oracle/np_image.py run_program on %d images, resize on 1" % n,
                      "hip_equals_port_on_the_sample": bool(all(np.array_equal(out[i].cpu().numpy(), want[i]) for i in range(n))
                                                            and np.array_equal(small[0].cpu().numpy(), want_small))}
    return leg


@_guard
def evaluator_leg(dev, with_cpu=True):
    """Evaluator.match_predictions (the reference's Python loop over every prediction of every class) with the
    matching on the GPU vs the NumPy port, on a VOC2007-test-sized synthetic problem (4952 images, 20 classes)."""
    from oracle import np_oracle as orc
    from ssd_keras_amd.eval_utils.average_precision_evaluator import Evaluator
    rng = np.random.RandomState(0)
    n_images, n_classes = 4952, 20
    labels, neutral, image_ids = [], [], []
    preds = [[] for _ in range(n_classes + 1)]
    for i in range(n_images):
        image_ids.append("%06d" % i)
        g = int(rng.randint(1, 6))
        cls = rng.randint(1, n_classes + 1, size=g)
        x0, y0 = rng.randint(0, 400, size=g), rng.randint(0, 300, size=g)
        lab = np.stack([cls, x0, y0, x0 + rng.randint(8, 120, size=g), y0 + rng.randint(8, 120, size=g)], axis=1).astype(np.int64)
        labels.append(lab)
        neutral.append(rng.uniform(size=g) < 0.15)
        for b in lab:
            for _ in range(int(rng.randint(0, 4))):
                j = rng.normal(0, 5, size=4)
                preds[int(b[0])].append((image_ids[-1], float(rng.uniform(0.01, 1)), float(b[1] + j[0]), float(b[2] + j[1]),
                                         float(b[3] + j[2]), float(b[4] + j[3])))
        for _ in range(int(rng.randint(0, 30))):
            c = int(rng.randint(1, n_classes + 1))
            x, y = rng.uniform(0, 450, size=2)
            preds[c].append((image_ids[-1], float(rng.uniform(0.01, 0.5)), float(x), float(y), float(x + rng.uniform(4, 90)), float(y + rng.uniform(4, 90))))
    gen = type("Gen", (), {})()
    gen.labels, gen.eval_neutral, gen.image_ids = labels, neutral, image_ids
    ev = Evaluator(model=None, n_classes=n_classes, data_generator=gen)
    ev.prediction_results = preds
    n_pred = sum(len(q) for q in preds)
    with torch.cuda.device(dev):
        ev.match_predictions(verbose=False)                      # (library load, workspace)
        ev.forget_packed_inputs()
        torch.cuda.synchronize()
        t = time.perf_counter()
        ev.match_predictions(verbose=False)                      # first evaluation of these lists: packs them (host-side tuple walking) + matches
        torch.cuda.synchronize()
        first_ms = 1e3 * (time.perf_counter() - t)
        t = time.perf_counter()
        tp, fp, _, _ = ev.match_predictions(verbose=False, ret=True)     # the same results again (another threshold / border mode would cost the same)
        torch.cuda.synchronize()
        gpu_ms = 1e3 * (time.perf_counter() - t)
    ev.get_num_gt_per_class(verbose=False)
    ev.compute_precision_recall(verbose=False)
    ev.compute_average_precisions(verbose=False)
    out = {"workload": "Evaluator.match_predictions, %d images, %d classes, %d predictions (synthetic, VOC2007-test sized)" % (n_images, n_classes, n_pred),
           "gpu_ms_total": round(gpu_ms, 2), "gpu_ms_first_call_incl_packing": round(first_ms, 2),
           "note": "wall time of Evaluator.match_predictions: all classes in ONE ssdhip_match_predictions_multi call + one download; the "
                   "reference-format inputs (per-class Python lists of tuples, list of label arrays) are packed to the device once per "
                   "object -- `first_call` includes that walk over 94 k tuples, `gpu_ms_total` is every later evaluation of the same results",
           "mAP_of_the_synthetic_problem": round(float(ev.compute_mean_average_precision()), 4)}
    if with_cpu:
        sub = [[]] + [preds[c] if c <= 2 else [] for c in range(1, n_classes + 1)]      # two classes, scaled up
        t = time.perf_counter()
        wtp, _, _, _ = orc.evaluator_match_predictions(sub, labels, image_ids, neutral, n_classes)
        cpu_s = time.perf_counter() - t
        frac = (len(preds[1]) + len(preds[2])) / max(n_pred, 1)
        same = all(np.array_equal(tp[c], wtp[c]) for c in (1, 2))
        out["cpu"] = {"ms_total_extrapolated": round(1e3 * cpu_s / max(frac, 1e-9), 1), "cores": 1, "kind": "port",
                      "sample": "oracle evaluator_match_predictions (NumPy port of average_precision_evaluator.py:538-736) on classes 1-2 "
                                "(%.1f %% of the predictions), scaled to all classes" % (100 * frac),
                      "identical_flags_on_the_sample": bool(same), "speedup": round((cpu_s / max(frac, 1e-9)) / (gpu_ms * 1e-3), 1)}
    return out


def cpu_decode_all_cores(y_host, kw, timeout_s=150):
    """The all-core CPU figure beside cpu_baseline's single-core one: tools/cpu_decode_all_cores.py in its own process (this
    one holds the HIP runtime and must not fork).  Never raises: a bench leg, not the metric."""
    import json
    import subprocess
    import sys
    import tempfile
    try:
        root = os.path.dirname(os.path.abspath(__file__))
        with tempfile.TemporaryDirectory() as d:
            np.save(os.path.join(d, "y.npy"), np.ascontiguousarray(y_host))
            with open(os.path.join(d, "kw.json"), "w") as f:
                json.dump(kw, f)
            run = subprocess.run([sys.executable, os.path.join(root, "tools", "cpu_decode_all_cores.py"), d], capture_output=True,
                                 text=True, timeout=timeout_s)
        if run.returncode != 0:
            return {"error": run.stderr.strip()[-300:]}
        return json.loads(run.stdout.strip().splitlines()[-1])
    except Exception as exc:                                                       # noqa: BLE001
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
