#!/usr/bin/env python3
"""Per-layer timing of the SSD300 VGG trunk convolutions at batch 32: libssdhip's fused implicit-GEMM kernel vs
MIOpen (F.conv2d) + the fused bias/ReLU pass.  GPU box only."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_keras_amd import _native as nat          # noqa: E402

LAYERS = [("head1", 19, 1024, 192, 3, 1), ("head3", 5, 256, 192, 3, 1), ("conv1_2", 300, 64, 64, 3, 1), ("conv2_1", 150, 64, 128, 3, 1), ("conv2_2", 150, 128, 128, 3, 1),
          ("conv3_1", 75, 128, 256, 3, 1), ("conv3_2", 75, 256, 256, 3, 1), ("conv4_1", 38, 256, 512, 3, 1),
          ("conv4_2", 38, 512, 512, 3, 1), ("conv5_1", 19, 512, 512, 3, 1), ("fc6", 19, 512, 1024, 3, 6),
          ("fc7", 19, 1024, 1024, 1, 1)]


def ev_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps


def main():
    B = int(os.environ.get("B", "32"))
    torch.backends.cudnn.benchmark = os.environ.get("FIND", "1") == "1"
    only = os.environ.get("ONLY")
    res = []
    for name, hw, cin, cout, k, dil in LAYERS:
        if only and name not in only.split(","):
            continue
        x = torch.randn((B, hw, hw, cin), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
        w = (torch.randn((cout, k, k, cin), device="cuda") / (k * k * cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        b = torch.randn((cout,), device="cuda").to(torch.bfloat16)
        flop = 2.0 * B * hw * hw * cin * cout * k * k
        t_ours = ev_ms(lambda: nat.conv2d_same(x, w, b, dilation=dil, relu=True))
        t_v1 = ev_ms(lambda: nat.conv2d_same(x, w, b, dilation=dil, relu=True, variant=1)) if os.environ.get("V1", "1") == "1" else float("nan")
        t_v4 = ev_ms(lambda: nat.conv2d_same(x, w, b, dilation=dil, relu=True, variant=4))
        t_v5 = ev_ms(lambda: nat.conv2d_same(x, w, b, dilation=dil, relu=True, variant=5))
        t_v6 = ev_ms(lambda: nat.conv2d_same(x, w, b, dilation=dil, relu=True, variant=6))
        t_v7 = ev_ms(lambda: nat.conv2d_same(x, w, b, dilation=dil, relu=True, variant=6))
        t_pool = t_unf = float("nan")
        if name in ("conv1_2", "conv2_2", "conv3_2"):
            t_pool = ev_ms(lambda: nat.conv2d_same_pool2(x, w, b, dilation=dil, relu=True))
            t_unf = ev_ms(lambda: nat.bias_act_maxpool(nat.conv2d_same(x, w, b, dilation=dil, relu=True), None, 2, 2, 0, True, relu=False))
        t_c64 = t_c64p = float("nan")
        if cin == 64 and k == 3:
            t_c64 = ev_ms(lambda: nat.conv3x3_c64(x, w, b, relu=True, pool=False))
            t_c64p = ev_ms(lambda: nat.conv3x3_c64(x, w, b, relu=True, pool=True))
        if os.environ.get("MIOPEN", "1") == "1":
            t_mi = ev_ms(lambda: F.conv2d(x, w, None, 1, dil * (k // 2), dil))
            t_mi_full = ev_ms(lambda: nat.bias_act(F.conv2d(x, w, None, 1, dil * (k // 2), dil), b, relu=True))
        else:
            t_mi = t_mi_full = float("nan")
        r = {"layer": name, "ours_us": round(t_ours * 1e3, 1), "ours_TFs": round(flop / t_ours / 1e9, 1), "v1_us": round(t_v1 * 1e3, 1), "v4_us": round(t_v4 * 1e3, 1), "v4_TFs": round(flop / t_v4 / 1e9, 1),
             "c64_us": round(t_c64 * 1e3, 1), "c64_pool_us": round(t_c64p * 1e3, 1), "conv_pool_fused_us": round(t_pool * 1e3, 1), "conv_then_pool_us": round(t_unf * 1e3, 1), "v5_us": round(t_v5 * 1e3, 1), "v5_TFs": round(flop / t_v5 / 1e9, 1), "v6_us": round(t_v6 * 1e3, 1), "v9_8waves_us": round(t_v7 * 1e3, 1), "v6_TFs": round(flop / t_v6 / 1e9, 1),
             "miopen_conv_us": round(t_mi * 1e3, 1), "miopen_TFs": round(flop / t_mi / 1e9, 1),
             "miopen_plus_epilogue_us": round(t_mi_full * 1e3, 1)}
        print(json.dumps(r), flush=True)
        res.append(r)
    # the strided / 'valid' extra layers: libssdhip's general entry vs MIOpen + the bias/ReLU pass
    for name, hw, cin, cout, stride, pad in (("conv6_2", 19, 256, 512, 2, 1), ("conv7_2", 10, 128, 256, 2, 1),
                                             ("conv8_2", 5, 128, 256, 1, 0), ("conv9_2", 3, 128, 256, 1, 0)):
        if only and name not in only.split(","):
            continue
        x = torch.randn((B, hw, hw, cin), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
        w = (torch.randn((cout, 3, 3, cin), device="cuda") / (9 * cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        b = torch.randn((cout,), device="cuda").to(torch.bfloat16)
        ho = (hw + 2 * pad - 3) // stride + 1
        flop = 2.0 * B * ho * ho * cin * cout * 9
        t_ours = ev_ms(lambda: nat.conv2d(x, w, b, stride=stride, padding=pad, relu=True))
        t_mi_full = ev_ms(lambda: nat.bias_act(F.conv2d(x, w, None, stride, pad, 1), b, relu=True))
        r = {"layer": name, "ours_us": round(t_ours * 1e3, 1), "ours_TFs": round(flop / t_ours / 1e9, 1),
             "miopen_plus_epilogue_us": round(t_mi_full * 1e3, 1)}
        print(json.dumps(r), flush=True)
        res.append(r)
    x = torch.randn((B, 300, 300, 3), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn((64, 3, 3, 3), device="cuda") / 27 ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    b = torch.randn((64,), device="cuda").to(torch.bfloat16)
    t1 = ev_ms(lambda: nat.conv3x3_cin3(x, w, b, relu=True))
    r = {"layer": "conv1_1", "ours_us": round(t1 * 1e3, 1), "write_GBps": round(B * 300 * 300 * 64 * 2 / t1 / 1e6, 1)}
    print(json.dumps(r), flush=True)
    res.append(r)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_conv.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
