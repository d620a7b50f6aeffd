#!/usr/bin/env python3
"""bench.py -- headline measurement: images/sec of SSD300 (VGG-16, 21 classes) forward + in-graph
decode (DecodeDetections: threshold 0.01, NMS 0.45, top-200) at batch 32 per GPU, synthetic
300x300x3 batches resident in HBM, random-init weights (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32]

N > 1: under a launcher (torch.distributed.run sets WORLD_SIZE; the driver's way) this process is one rank; WITHOUT one,
`python bench.py --gpus N` spawns the N ranks itself (torch.distributed.run, 127.0.0.1 rendezvous, one rank per GPU,
RCCL) and asserts WORLD_SIZE == N either way.  The data path has no collective (every rank decodes its own 32 images:
weak scaling), the barrier + max-over-ranks timing does; `rccl_ranks_seen` on the line is an all-reduce of ones.
Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      the dominant hand-written kernel of the decode path against HBM bandwidth: algorithmic bytes of
                the decode path per launch (N*(C+12)*4 read + top_k*6*4 written per image) / that
                kernel's average launch duration, measured with events on the launch stream;
  cpu_baseline  the NumPy port of the reference decoder (oracle/np_oracle.py) timed on this box's host cores on a
                bounded sample of the same predictions (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
SSD300_FWD_GFLOP_PER_IMG = 62.747


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", type=int, default=0,
                    help="0 (default): decode on the forward's stream; 1: on a second stream beside the next forward -- measured 3 %% "
                         "SLOWER since the convolutions fill the CUs (the persistent conv64 kernel owns every CU's LDS)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="images for the CPU baseline (0: auto, ~10-30 s)")
    ap.add_argument("--graph", type=int, default=1, help="1: the timed step is one HIP-graph launch (model.graphed); 0: eager launches")
    ap.add_argument("--graph-warmup", type=int, default=40,
                    help="untimed replays of the captured step graph before the timed region (beside the W eager warm-up steps): the chip "
                         "needs ~12 steps (25 ms) behind the capture's idle time to reach its steady clock -- profiles/r06zs_*")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs (encoder / loss / sparse decode / training step)")
    ap.add_argument("--train-steps", type=int, default=6, help="timed steps of the training-step leg (0: skip it)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="create the --gpus N ranks, bring up the process group (RCCL on GPUs, gloo where there is none), all-reduce "
                         "a one per rank, print {'ranks_seen': N, ...} and exit: the launcher path without any measurement")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-spawned ranks (0: a free one)")
    return ap.parse_args()


def self_spawn(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: re-execute this script as N ranks under
    torch.distributed.run (one process per GPU, 127.0.0.1 rendezvous) -- the command the driver itself uses -- and return its exit
    code.  With a launcher environment (WORLD_SIZE set) this is never called: the process IS one of the ranks."""
    import socket
    import subprocess
    if (torch.cuda.is_available() and torch.cuda.device_count() < args.gpus and os.environ.get("SSD_BENCH_SHARE_GPU", "0") != "1"):
        print("bench.py: --gpus %d but this box has %d GPU(s) (one rank per GPU; SSD_BENCH_SHARE_GPU=1 SSD_BENCH_BACKEND=gloo puts every "
              "rank on GPU 0 to exercise the launch path only)" % (args.gpus, torch.cuda.device_count()), file=sys.stderr)
        return 2
    port = args.master_port
    if not port:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_launch(args):
    """The N-rank launch path and nothing else: process group up, an all-reduce of ones, one JSON line from rank 0."""
    from ssd_keras_amd import distributed as dp
    has_gpu = torch.cuda.is_available()
    backend = os.environ.get("SSD_BENCH_BACKEND", "nccl" if has_gpu else "gloo")
    if has_gpu and os.environ.get("SSD_BENCH_SHARE_GPU", "0") == "1":
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local_rank = dp.init_from_env(backend)
    assert world == args.gpus, "launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    dev = torch.device("cuda", local_rank) if (has_gpu and backend == "nccl") else torch.device("cpu")
    seen = int(dp.sum_over_ranks(1.0, device=dev))
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": args.gpus, "world_size": world, "ranks_seen": seen, "backend": backend,
                          "device": str(dev)}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if seen == args.gpus else 1


def _port_ratio():
    """(range of port time / reference time, where it comes from): REPLAYED from the newest committed measurement made in the build
    container (tools/port_vs_reference_time.py; the GPU box has no /root/reference), never measured by this run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*port_vs_reference_time_ratio.json")))
    if not files:
        return None, "not measurable on the GPU box (no /root/reference)"
    try:
        rec = json.load(open(files[-1]))
        return rec["port_over_reference_range"], ("replayed from profiles/%s (the port against the REAL reference on the same arrays, build "
                                                  "container, one core; the port is the FASTER of the two: the CPU baseline is conservative)"
                                                  % os.path.basename(files[-1]))
    except Exception:                                        # noqa: BLE001
        return None, "profiles/*port_vs_reference_time_ratio.json unreadable"


def conv_choice_label(key):
    """Readable name of an autotune key of ssd_keras_amd.models._common.SSDModel._pick (never raises)."""
    try:
        kind = key[0]
        if kind == "head":                                   # ("head", layer, x shape, conf Cout, loc Cout)
            return "head%d %s -> %d+%d k3" % (key[1], "x".join(str(v) for v in key[2]), key[3], key[4])
        if kind == "heads":                                  # ("heads", shapes of all source maps, classes)
            return "all heads (%d source maps)" % len(key[1])
        label = "%s %s -> %s k%s d%s" % (kind, "x".join(str(v) for v in key[1]), key[2], key[3], key[4])
        if kind == "act" and len(key) >= 8 and (key[6] != 1 or key[7] != key[4] * (key[3] // 2)):
            label += " s%s p%s" % (key[6], key[7])              # strided / partially padded extra layers
        return label
    except Exception:
        return repr(key)


def event_ms(fn, reps):
    """Average milliseconds of `fn()` over `reps` back-to-back launches, events on the current stream."""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / reps


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))                   # no launcher around us: become the launcher of N ranks
    if args.dry_launch:
        sys.exit(dry_launch(args))
    # one process per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher; the process group (RCCL) comes up through the
    # package's own helper (device selected before init, dmabuf IPC mode, 127.0.0.1 rendezvous)
    from ssd_keras_amd import distributed as dp
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # (SSD_BENCH_BACKEND=gloo SSD_BENCH_SHARE_GPU=1: every rank on GPU 0 over gloo -- only to exercise the N > 1 code path on a
    # one-GPU box, tools/gpu_two_ranks_one_gpu.sh; the numbers of such a run mean nothing)
    backend = os.environ.get("SSD_BENCH_BACKEND", "nccl")
    if os.environ.get("SSD_BENCH_SHARE_GPU", "0") == "1":
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local_rank = dp.init_from_env(backend)
    assert world == args.gpus, ("--gpus %d but the launcher created WORLD_SIZE=%d ranks: n_gpus on the result line would lie"
                                % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # every rank adds a one over the bench's backend (RCCL unless overridden): the line carries how many ranks really took part
    ranks_seen = int(dp.sum_over_ranks(1.0, device=dev)) if world > 1 else 1

    from ssd_keras_amd import _native as nat
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.models.keras_ssd300 import ssd_300
    nat.load()                                    # fail loudly when the HIP library is missing

    B = args.batch
    torch.manual_seed(1234 + rank)
    cfg = syn.SSD300_VOC
    model = ssd_300((300, 300, 3), cfg["n_classes"], mode="inference", scales=cfg["scales"],
                    aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"],
                    confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400).to(dev)
    model = model.to(memory_format=torch.channels_last).eval()
    if args.dtype == "bf16":
        model = model.to(torch.bfloat16)
    torch.backends.cudnn.benchmark = True
    rng = np.random.RandomState(rank)
    images = torch.from_numpy(rng.randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).to(dev)

    def step():
        with torch.no_grad():
            return model(images)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # ---- timed region: exactly K steps ---------------------------------------------------------
    # A step = forward + DecodeDetections of ITS predictions, on one stream (default).  With --overlap 1 the decode is
    # enqueued on a second HIP stream behind an event so that the NMS of step i runs beside the convolutions of step i+1;
    # every step's decode finishes inside the timed region either way.
    dec_stream = torch.cuda.Stream(device=dev) if args.overlap else torch.cuda.current_stream(dev)
    dec_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # The step as ONE HIP-graph launch (the same kernels on the same streams, recorded once): the eager step issues ~45 launches
    # from Python, and on a busy host that alone can exceed the 3 ms of GPU work.  --graph 0, --overlap 1 or a failed capture: eager.
    runner, launch = None, "eager (one Python-issued launch per kernel)"
    if args.graph and not args.overlap:
        try:
            runner = model.graphed(images)
            for _ in range(max(3, args.graph_warmup)):
                out = runner(images)
            torch.cuda.synchronize()
            launch = "hip_graph (forward + DecodeDetections captured once, replayed per step)"
        except Exception as exc:                     # noqa: BLE001 -- reported, and the eager step is the same work
            runner = None
            launch = "eager (graph capture failed: %s)" % (repr(exc)[:160])
            torch.cuda.synchronize()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g0.record()
    for i in range(args.steps):
        with torch.no_grad():
            if runner is not None:
                out = runner(images)
                continue
            if not args.overlap:
                out = model(images)                  # forward + DecodeDetections straight from the head outputs (no y_pred in HBM)
                continue
            pred = model.raw_predictions(images)
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(dec_stream):
                dec_stream.wait_event(ready)
                dec_ev[i][0].record()
                out = model.decoder(pred)
                dec_ev[i][1].record()
            pred.record_stream(dec_stream)
    g1.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms_per_step = g0.elapsed_time(g1) / args.steps   # device-side span of the same K steps (< wall clock when the host is the limit)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    if args.overlap:
        decode_ms_in_step = float(np.mean([a.elapsed_time(b) for a, b in dec_ev]))
    else:
        # the decode of the step as the step runs it (DecodeDetections.forward_from_heads: scan_heads_kernel + nms + topk): a few
        # extra steps with the layer's event hook on, outside the timed region
        model.decoder.timing_events = []
        with torch.no_grad():
            for _ in range(10):
                model(images)
        torch.cuda.synchronize()
        in_step = [a.elapsed_time(b) for a, b in model.decoder.timing_events]
        decode_ms_in_step = float(np.median(in_step))                 # (median: a first call can carry a one-off allocation)
        model.decoder.timing_events = None
    step_out = out                                   # (B, top_k, 6) float32: what the timed steps returned
    with torch.no_grad():
        pred = model.raw_predictions(images)         # the step's own prediction tensor: input of the per-kernel timing and the CPU baseline

    # ---- per-kernel timing of the decode path on the step's own predictions ----------------------
    N, C = pred.shape[1], pred.shape[2] - 12
    algo_bytes = B * (N * (C + 12) * 4 + 200 * 6 * 4)
    d = model.decoder
    dkw = dict(conf_thresh=d.confidence_thresh, iou_thresh=d.iou_threshold, top_k=d.top_k, nms_cap=d.nms_max_output_size,
               class_agnostic=False, semantics=nat.SEM_KERAS, coords="centroids", normalize_coords=True,
               img_height=300, img_width=300, border_pixels="half", out_dtype=nat.F32, out_rows=d.top_k)
    outs = nat.decode(pred, **dkw)
    reps = 50
    stage_ms = {}
    for name, mask in (("scan_kernel", 1), ("nms_kernel", 2), ("topk_kernel", 4), ("decode_path", 7)):
        nat.decode(pred, stages=mask, outputs=outs, **dkw)
        torch.cuda.synchronize()
        stage_ms[name] = event_ms(lambda m=mask: nat.decode(pred, stages=m, outputs=outs, **dkw), reps)
    with torch.no_grad():
        fwd_eager_ms = event_ms(lambda: model.raw_predictions(images), 10)
    # The convolution stack AS THE TIMED STEP RUNS IT: a HIP graph of model.head_outputs -- input Lambdas, trunk, extra layers, packed
    # predictor heads on the step's two streams, nothing behind them -- replayed on its capture stream behind the same untimed replays as
    # the step.  (Rounds 1-6 reported the EAGER model.raw_predictions here: ~45 launches issued from Python with their gaps, plus the
    # prediction assembly the step never runs, measured without a warm-up: 2.02 ms on a box whose whole graph step took 1.95.)
    fwd_ms, fwd_how = fwd_eager_ms, "eager model.raw_predictions (forward + prediction assembly, launch gaps included)"
    if runner is not None:
        try:
            fwd_runner = model.graphed(images, heads_only=True)
            with torch.cuda.stream(fwd_runner.stream), torch.no_grad():
                for _ in range(max(3, args.graph_warmup)):
                    fwd_runner(images)
                torch.cuda.synchronize()
                a_ev, b_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_ev.record()
                for _ in range(args.steps):
                    fwd_runner(images)
                b_ev.record()
                b_ev.synchronize()
            fwd_ms = a_ev.elapsed_time(b_ev) / args.steps
            fwd_how = "hip_graph of model.head_outputs (the step's convolution stack on its two streams, no decode), %d replays" % args.steps
        except Exception as exc:                     # noqa: BLE001 -- the eager number stands
            fwd_how += "; graph of head_outputs failed: %s" % (repr(exc)[:120])
            torch.cuda.synchronize()
    dom = max(("scan_kernel", "nms_kernel", "topk_kernel"), key=lambda k: stage_ms[k])
    full_name = {"scan_kernel": "scan_kernel", "nms_kernel": "nms_kernel<POL_TF32, 512, false> (+ the redo launch of nms_kernel<POL_TF32, 512, true>: idle workgroups)", "topk_kernel": "topk_kernel<float>"}
    achieved = algo_bytes / (stage_ms[dom] * 1e-3) / 1e9
    traffic = None                                   # HBM bytes per launch of the dominant kernel from the committed PMC passes
    import glob
    profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*decode_pmc_traffic.json")))
    if profs:
        try:
            allrec = json.load(open(profs[-1]))
            # (K4 = the 512-thread launch + the redo launch of the full kernel, both counted; one record for every other kernel)
            recs = [allrec[k] for k in sorted(allrec) if k == dom or k.startswith(dom + "<")]
            traffic = {"hbm_bytes_per_launch": round(sum(r["hbm_bytes"] for r in recs)),
                       "read": round(sum(r["hbm_read_bytes"] for r in recs)),
                       "write": round(sum(r["hbm_write_bytes"] for r in recs)),
                       "ratio_to_algorithmic_bytes": round(sum(r["hbm_bytes"] for r in recs) / algo_bytes, 4),
                       "source": "profiles/" + os.path.basename(profs[-1]),
                       "measured_in_this_run": False,
                       "note": "REPLAYED from the newest committed PMC profile (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                               "passes, tools/gpu.sh pmc_decode): counters cannot be read inside this process"}
        except Exception:
            traffic = None
    roofline = {"kernel": full_name[dom], "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms": {k: round(v, 5) for k, v in stage_ms.items()},
                "decode_ms_in_step": round(decode_ms_in_step, 5),
                "note": "kernel_ms times the three kernels on the assembled prediction tensor; the step itself decodes straight from "
                        "the head outputs (scan_heads_kernel instead of scan_kernel): decode_ms_in_step, events around that call",
                "decode_path_GBps": round(algo_bytes / (stage_ms["decode_path"] * 1e-3) / 1e9, 2)}
    conv_tflops = B * SSD300_FWD_GFLOP_PER_IMG / 1e3 / (fwd_ms * 1e-3)
    from ssd_keras_amd.models._common import SSDModel
    choices = {}
    for key, name in SSDModel._conv_choice.items():          # which kernel the per-shape autotune kept for each convolution
        choices[conv_choice_label(key)] = name
    # the chip does not hold its nominal clock with every SIMD issuing MFMAs: 2.39 GHz with one workgroup resident, 1.75-1.90 GHz under a full
    # MFMA load (tools/micro/memtime_vs_mfma.hip, profiles/r04q2_clock64_is_shader_cycles_and_full_load_clock.txt) -- the fraction against the
    # peak THAT clock allows (midpoint 1.825 GHz) beside the one against the nominal 2.5 PF/s
    sustained_peak = MFMA_PEAK_TFLOPS[args.dtype] * (1.825 / 2.39) if args.dtype == "bf16" else None
    conv = {"bound": "mfma", "forward_ms": round(fwd_ms, 4), "forward_ms_measured_as": fwd_how,
            "forward_ms_eager_raw_predictions": round(fwd_eager_ms, 4), "achieved": round(conv_tflops, 2),
            "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s", "frac": round(conv_tflops / MFMA_PEAK_TFLOPS[args.dtype], 5),
            "frac_at_sustained_clock": round(conv_tflops / sustained_peak, 5) if sustained_peak else None,
            "sustained_clock_note": "peak x 1.825 / 2.39 GHz: the shader clock under a full MFMA load, REPLAYED from "
                                    "profiles/r04q2_clock64_is_shader_cycles_and_full_load_clock.txt (1.75-1.90 GHz), not measured by this run",
            "note": "per layer shape the fastest of libssdhip's MFMA kernels, timed once (kernel_per_layer): the slab kernel "
                    "(csrc/ssdhip_convh.hip: halo / halo_pool / halo_grouped), the fused conv1_1 + conv1_2 + pool1 kernel "
                    "(conv1_block), the resident-weight kernel (c64), the implicit-GEMM kernels (igemm*); %s; 62.747 GFLOP/img "
                    "(SURVEY App. B); per-kernel MFMA-busy counters of this forward: profiles/*_pmc_mfma_per_kernel.txt" % args.dtype,
            "kernel_per_layer": choices}

    # ---- CPU baseline: NumPy port of the reference decoder on a bounded sample (rank 0, N=1) -----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import np_oracle as orc
        from oracle import parity as par
        y_host = pred.float().cpu().numpy()
        kw = dict(confidence_thresh=0.01, iou_threshold=0.45, top_k=200, normalize_coords=True, img_height=300, img_width=300)
        kw_all = dict(kw, top_k="all")
        with np.errstate(all="ignore"):              # random-init offsets overflow exp(): inf / NaN boxes are part of this workload
            t = time.perf_counter()
            orc.decode_detections(y_host[:1], **kw)
            one = time.perf_counter() - t
            n_img = args.cpu_sample or int(max(1, min(B, round(12.0 / max(one, 1e-3)))))
            orc.NMS_WORK.update(iou_pairs=0, kept=0)
            # timed: the port exactly as the reference runs it (top_k = 200; "det": the exp shared with the kernel, bit-comparable)
            t = time.perf_counter()
            ref = orc.decode_detections(y_host[:n_img], exp_mode="det", **kw)
            cpu_s = time.perf_counter() - t
            pairs_per_img = orc.NMS_WORK["iou_pairs"] / n_img            # box pairs the reference's NMS formulation evaluates
        cpu = {"value": round(n_img / cpu_s, 4), "unit": "images/sec (decode_detections only)", "cores": 1, "kind": "port",
               "ms_per_img": round(1e3 * cpu_s / n_img, 3),
               "sample": "oracle/np_oracle.decode_detections (NumPy port of ssd_output_decoder.py:111-226) on the first %d of "
                         "%d images of the step's own predictions (random-init weights: every anchor passes 0.01 -> dense "
                         "regime; He-init on 0..255 inputs saturates the softmax and overflows exp() -- BASELINE's named config, "
                         "degenerate for a detector: `decode_sparse` below is the trained-model-like regime); the forward pass has "
                         "no CPU reference here (TensorFlow absent)" % (n_img, B),
               # the port's time / the real reference's time on the same arrays: only measurable where /root/reference imports (the
               # build container; 0.74-0.78 there) -- never on the GPU box, so not restated here as a measurement
               "port_vs_reference_time_ratio": _port_ratio()[0],
               "port_vs_reference_time_ratio_source": _port_ratio()[1],
               "gpu_decode_ms_per_img": round(stage_ms["decode_path"] / B, 5),
               "speedup_decode": round((cpu_s / n_img) / (stage_ms["decode_path"] * 1e-3 / B), 1),
               "host_cpus": os.cpu_count(),
               # SURVEY 8d secondary figure (the dense regime is VALU-bound): IoU evaluations of the reference's formulation
               # (every kept box against everything still alive) per second, CPU port vs the HIP NMS kernel on the same images
               "nms_iou_pairs_per_img": round(pairs_per_img), "cpu_iou_pairs_per_sec": round(pairs_per_img * n_img / cpu_s),
               "gpu_iou_pairs_per_sec": round(pairs_per_img * B / (stage_ms["nms_kernel"] * 1e-3))}
        # ---- parity at the bench's own scale (flags, never the metric) ----
        # (a) the timed step's own output (DecodeDetections layer semantics: tf.nn.top_k breaks ties deterministically) against
        #     the layer restatement on the sampled images: exact comparison
        try:
            with np.errstate(all="ignore"):
                want_l = orc.decode_detections_layer(y_host[:n_img], nms_max_output_size=400, exp_mode="det", **kw)
            cpu["step_output_vs_layer_oracle"] = par.layer_parity(step_out[:n_img].float().cpu().numpy(), want_l)
        except Exception as exc:                                        # noqa: BLE001
            cpu["step_output_vs_layer_oracle"] = "error: %s: %s" % (type(exc).__name__, exc)
        # (b) the HIP decode_detections (NumPy semantics) against the port, tie-aware: np.argpartition's choice among equal
        #     confidences at the 200-row cut is arbitrary in the reference itself (oracle/parity.py)
        try:
            from ssd_keras_amd.ssd_encoder_decoder.ssd_output_decoder import decode_detections as hip_decode
            with np.errstate(all="ignore"):
                ref_all = orc.decode_detections(y_host[:n_img], exp_mode="det", **kw_all)
            got = hip_decode(pred[:n_img], **kw)
            got_all = hip_decode(pred[:n_img], **kw_all)
            cpu["hip_vs_port_on_the_sample"] = par.decode_parity(got, ref_all, 200, got_all)
            cpu["hip_vs_port_on_the_sample"]["port_top200_consistent_with_its_uncut_set"] = par.decode_parity(ref, ref_all, 200)["ok"]
            cpu["detections_on_the_sample"] = int(sum(w.shape[0] for w in ref if w.size))
        except Exception as exc:                                        # noqa: BLE001 -- a reported flag, never the metric
            cpu["hip_vs_port_on_the_sample"] = "error: %s: %s" % (type(exc).__name__, exc)
        # the all-core figure beside the single-core one (own process, one batch item per task)
        import bench_extra as bx_cpu
        cpu["all_cores"] = bx_cpu.cpu_decode_all_cores(y_host, kw)

    # ---- the same step on a NON-saturated prediction distribution: the predictor heads tamed in place (filters
    #      x 1e-2, background bias + 4 -- the training leg's initialisation), so that the in-step decode is also timed where confidences
    #      are distinct instead of ~400 rows tying at 1.0 with inf / NaN boxes.  Same graph, same kernels; reported beside `value`. -------
    tamed = None
    if rank == 0 and world == 1:
        try:
            with torch.no_grad():
                for head in model.conf_heads:
                    head.weight.mul_(1e-2)
                    head.bias.view(-1, int(C))[:, 0] = 4.0
                for head in model.loc_heads:
                    head.weight.mul_(1e-2)
                run = (lambda: runner(images)) if runner is not None else (lambda: model(images))
                import contextlib
                # (the step graph is the OLDER of the model's two graphs now: on its capture stream it replays without the guard's waits)
                with (torch.cuda.stream(runner.stream) if runner is not None else contextlib.nullcontext()):
                    for _ in range(max(3, args.graph_warmup) if runner is not None else 3):
                        out_t = run()
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    for _ in range(args.steps):
                        out_t = run()
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t
                model.decoder.timing_events = []
                for _ in range(10):
                    model(images)
                torch.cuda.synchronize()
                dec_t = float(np.median([a.elapsed_time(b) for a, b in model.decoder.timing_events]))
                model.decoder.timing_events = None
                rows = out_t.float()
                kept = (rows[:, :, 1] > 0).sum(dim=1).float()
                top = rows[:, 0, 1]
            tamed = {"value": round(B * args.steps / dt, 2), "unit": "images/sec", "ms_per_step": round(1e3 * dt / args.steps, 4),
                     "decode_ms_in_step": round(dec_t, 5), "detections_per_image_mean": round(float(kept.mean()), 1),
                     "top_confidence_mean": round(float(top.mean()), 4), "finite_boxes": bool(torch.isfinite(rows[:, :, 2:]).all()),
                     "init": "predictor-head filters x 1e-2, background bias + 4 (bench_extra.train_leg's tamed heads)"}
        except Exception as exc:                                            # noqa: BLE001 -- a companion figure, never the metric
            tamed = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}

    # ---- secondary legs (bench_extra.py): encoder / loss / sparse decode vs the CPU port on rank 0 at N=1; the
    #      data-parallel training step (configs[2], configs[3]) on every rank.  None of them touches `value`. ------------
    extra = {}
    if not args.no_extra:
        import bench_extra as bx
        del model, out
        torch.cuda.empty_cache()
        with_cpu = world == 1 and not args.no_cpu_baseline
        if rank == 0:
            extra["encoder"] = bx.encoder_leg(dev, B, with_cpu)
            extra["loss"] = bx.loss_leg(dev, B, with_cpu)
            extra["decode_sparse"] = bx.sparse_decode_leg(dev, B, with_cpu)
            extra["ssd512_decode"] = bx.ssd512_decode_leg(dev, with_cpu)
            extra["conv_roofline_fp32"] = bx.fp32_forward_leg(dev, B)
            extra["conv_roofline_fp32x3"] = bx.fp32x3_forward_leg(dev, B, extra["conv_roofline_fp32"])
            extra["other_models_forward"] = bx.other_models_forward_leg(dev)
            extra["evaluator"] = bx.evaluator_leg(dev, with_cpu)
            extra["augmentation"] = bx.augmentation_leg(dev, B, with_cpu)
        if args.train_steps > 0:
            tr = bx.train_leg(dev, rank, world, B, steps=args.train_steps, warmup=3)
            if isinstance(tr, dict) and "miopen" in str(tr.get("error", "")).lower():
                # MIOpen's benchmark (find) mode intermittently rejects a problem with miopenStatusBadParm on this stack (seen on
                # the training step's framework convolutions): one retry on its default heuristics
                torch.backends.cudnn.benchmark = False
                torch.cuda.synchronize()
                tr = bx.train_leg(dev, rank, world, B, steps=args.train_steps, warmup=3)
                if isinstance(tr, dict):
                    tr["note_miopen"] = "retried with torch.backends.cudnn.benchmark = False after a miopenStatusBadParm in find mode"
            if rank == 0:
                extra["train_step"] = tr

    if rank == 0:
        ips = world * B * args.steps / elapsed
        line = {"metric": "images/sec SSD300 fwd+decode @batch32", "value": round(ips, 2), "unit": "images/sec",
                "n_gpus": world, "rccl_ranks_seen": ranks_seen, "collective_backend": backend if world > 1 else None,
                "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "data": "synthetic",
                "dtype": "f32 (decode, threshold, TF-style f32 NMS, top-k); %s conv backbone" % args.dtype,
                "config": {"workload": "SSD300 VGG-16 inference (mode='inference': forward + DecodeDetections), 21 classes, "
                                       "batch %d per GPU, conf 0.01 / NMS 0.45 / top-200, random-init weights, synthetic "
                                       "300x300x3 uint8-range images" % B,
                           "per_gpu_batch": B, "global_batch": world * B, "anchors": int(N), "classes": int(C),
                           "conv_dtype": args.dtype, "decode_dtype": "f32 decode + f32 IoU (DecodeDetections layer); decode_detections: f32 decode, f64 IoU", "parallelism": "replicas x%d" % world,
                           "launch": launch, "gpu_ms_per_step": round(gpu_ms_per_step, 4),
                           "untimed_graph_replays_before_the_timed_region": max(3, args.graph_warmup) if runner is not None else 0,
                           "decode_stream": "second HIP stream, overlapped with the next forward" if args.overlap else
                                            "same stream; DecodeDetections reads the head outputs directly (no y_pred in HBM)"},
                "value_tamed_heads": tamed, "roofline": roofline, "conv_roofline": conv, "cpu_baseline": cpu}
        line.update(extra)
        # ---- scalars at the top level (the driver's record keeps scalar keys only) ----
        def dig_any(obj, *path):
            for k in path:
                if not isinstance(obj, dict) or k not in obj:
                    return None
                obj = obj[k]
            return obj

        def dig(obj, *path):
            for k in path:
                if not isinstance(obj, dict) or k not in obj:
                    return None
                obj = obj[k]
            return obj if isinstance(obj, (int, float, bool)) else None
        flags = []

        def flag(v):
            if isinstance(v, dict) and ("ok" in v or "equal" in v):
                flags.append(bool(v.get("ok", v.get("equal"))))
            elif isinstance(v, bool):
                flags.append(v)
            elif v is not None:
                flags.append(False)                                      # an "error: ..." string is not green
        if isinstance(cpu, dict):
            flag(cpu.get("step_output_vs_layer_oracle"))
            flag(cpu.get("hip_vs_port_on_the_sample"))
        flag(dig_any(extra, "encoder", "cpu", "hip_matches_port"))
        flag(dig_any(extra, "loss", "cpu", "hip_loss_within_1e-4_of_port"))
        flag(dig_any(extra, "decode_sparse", "cpu", "hip_vs_port_on_the_sample"))
        flag(dig_any(extra, "ssd512_decode", "sparse_bias7_conf0.01", "cpu", "hip_vs_port_on_the_sample"))
        flag(dig_any(extra, "augmentation", "cpu", "hip_equals_port_on_the_sample"))
        flag(dig_any(extra, "augmentation", "augment_batch_equals_the_per_image_chain_under_each_seed"))
        flag(dig_any(extra, "augmentation", "augment_batch_global_stream_equals_the_per_image_chain"))
        scal = {"value_reference_precision": dig(extra, "conv_roofline_fp32x3", "images_per_sec"),
                "reference_precision_ms_per_step": dig(extra, "conv_roofline_fp32x3", "step_ms_fwd_plus_decode"),
                "value_tamed_heads_img_s": dig(tamed, "value"), "decode_ms_in_step": round(decode_ms_in_step, 5),
                "decode_ms_in_step_tamed": dig(tamed, "decode_ms_in_step"), "conv_frac": conv["frac"], "conv_frac_at_sustained_clock": conv["frac_at_sustained_clock"], "forward_ms": conv["forward_ms"],
                "nms_kernel_us": round(1e3 * stage_ms["nms_kernel"], 2), "scan_kernel_us": round(1e3 * stage_ms["scan_kernel"], 2),
                "nms_traffic_ratio_replayed_from_profile": (traffic or {}).get("ratio_to_algorithmic_bytes") if dom == "nms_kernel" else None,
                "train_step_ms": dig(extra, "train_step", "ms_per_step"), "train_images_per_sec": dig(extra, "train_step", "images_per_sec"),
                "loss_forward_ms": dig(extra, "loss", "fwd_ms"), "encoder_kernels_ms": dig(extra, "encoder", "gpu_ms_per_batch_kernels"),
                "augment_batch_img_s": dig(extra, "augmentation", "augment_batch_images_per_sec"),
                "augment_batch_global_stream_img_s": dig(extra, "augmentation", "augment_batch_global_stream_images_per_sec"),
                "evaluator_ms": dig(extra, "evaluator", "gpu_ms_total"),
                "ssd512_forward_img_s_batch8": dig(extra, "other_models_forward", "ssd512_voc_21_classes", "batch8", "images_per_sec"),
                "ssd7_forward_img_s_batch8": dig(extra, "other_models_forward", "ssd7_300x300_5_classes", "batch8", "images_per_sec"),
                "cpu_baseline_ms_per_img_1core": dig(cpu, "ms_per_img"),
                "parity_all_green": (all(flags) if flags else None), "parity_flags_counted": len(flags)}
        line.update({k: v for k, v in scal.items() if k not in line})
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
