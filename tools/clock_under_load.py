"""What the chip clocks at, and what it draws, while ONE kernel of the forward runs back to back for a few seconds: `rocm-smi`
sampled from a thread beside a loop of launches.  The bf16 MFMA peak (2.5 PF/s) is quoted at the nominal 2.4 GHz; every convolution
kernel here runs under the board's power limit at a lower clock -- this script measures by how much, per kernel.  GPU box."""
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

SECONDS = float(os.environ.get("LOAD_SECONDS", "4"))


def smi():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
    try:
        d = json.loads(out)
        card = d[sorted(d)[0]]
    except Exception:
        return {"raw": out[:400]}
    rec = {}
    for k, v in card.items():
        kl = k.lower()
        if "sclk" in kl:
            m = re.search(r"(\d+)\s*mhz", str(v).lower())
            rec["sclk_mhz"] = int(m.group(1)) if m else str(v)
        elif "mclk" in kl:
            m = re.search(r"(\d+)\s*mhz", str(v).lower())
            rec["mclk_mhz"] = int(m.group(1)) if m else str(v)
        elif "power" in kl and "(w)" in kl:
            try:
                rec["power_w"] = float(v)
            except Exception:
                rec["power_w"] = str(v)
        elif "temperature" in kl and "junction" in kl:
            rec["tj_c"] = v
    return rec


def under_load(name, fn, flop=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        time.sleep(0.8)                                   # let the clock settle
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.15)

    th = threading.Thread(target=sampler)
    th.start()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, n = time.perf_counter(), 0
    a.record()
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e.record()
    e.synchronize()
    stop.set()
    th.join()
    us = a.elapsed_time(e) / n * 1e3
    clk = [s["sclk_mhz"] for s in samples if isinstance(s.get("sclk_mhz"), int)]
    pw = [s["power_w"] for s in samples if isinstance(s.get("power_w"), float)]
    rec = {"kernel": name, "us_per_launch": round(us, 1), "samples": len(samples),
           "sclk_mhz_median": sorted(clk)[len(clk) // 2] if clk else None, "sclk_mhz_min_max": [min(clk), max(clk)] if clk else None,
           "power_w_median": sorted(pw)[len(pw) // 2] if pw else None, "power_w_max": max(pw) if pw else None}
    if flop:
        rec["tflops"] = round(flop / us / 1e6, 1)
        if clk:
            rec["frac_of_peak_at_that_clock"] = round(flop / us / 1e6 / (2500.0 * rec["sclk_mhz_median"] / 2400.0), 3)
    if not clk and samples:
        rec["first_sample"] = samples[0]
    print(json.dumps(rec), flush=True)


def bf(*shape, scale=1.0):
    return (torch.randn(shape, device="cuda") * scale).to(torch.bfloat16)


print(json.dumps({"idle": smi()}), flush=True)
B = 32
# conv2_1 on the Cin = 64 kernel
x = bf(B, 150, 150, 64).permute(0, 3, 1, 2); w = bf(128, 3, 3, 64, scale=1 / 24).permute(0, 3, 1, 2); b = bf(128)
under_load("conv2_1 (conv64_kernel, 64 -> 128, 150 x 150)", lambda: nat.conv3x3_c64(x, w, b, relu=True, pool=False), 2 * B * 150 * 150 * 576 * 128)
# conv4_2 on the slab kernel
x4 = bf(B, 38, 38, 512).permute(0, 3, 1, 2); w4 = bf(512, 3, 3, 512, scale=1 / 68).permute(0, 3, 1, 2); b4 = bf(512)
under_load("conv4_2 (convh_kernel, 512 -> 512, 38 x 38)", lambda: nat.conv2d_same(x4, w4, b4, dilation=1, relu=True), 2 * B * 38 * 38 * 4608 * 512)
# conv3_2 on the slab kernel
x3 = bf(B, 75, 75, 256).permute(0, 3, 1, 2); w3 = bf(256, 3, 3, 256, scale=1 / 48).permute(0, 3, 1, 2); b3 = bf(256)
under_load("conv3_2 (convh_kernel, 256 -> 256, 75 x 75)", lambda: nat.conv2d_same(x3, w3, b3, dilation=1, relu=True), 2 * B * 75 * 75 * 2304 * 256)
# fc6 on the image kernel
x6 = bf(B, 19, 19, 512).permute(0, 3, 1, 2); w6 = bf(1024, 3, 3, 512, scale=1 / 68).permute(0, 3, 1, 2); b6 = bf(1024)
under_load("fc6 (conv_image_kernel, 512 -> 1024, 19 x 19, dilation 6)", lambda: nat.conv2d_same(x6, w6, b6, dilation=6, relu=True), 2 * B * 19 * 19 * 4608 * 1024)
# the fused first block
x1 = bf(B, 300, 300, 3).permute(0, 3, 1, 2); w1 = bf(64, 3, 3, 3, scale=0.2).contiguous(memory_format=torch.channels_last); b1 = bf(64)
w2 = bf(64, 3, 3, 64, scale=1 / 24).permute(0, 3, 1, 2); b2 = bf(64)
under_load("conv1_1 + conv1_2 + pool1 (fused conv64_kernel)", lambda: nat.conv1_block(x1, w1, b1, w2, b2, relu=True, pool=True),
           2 * B * 300 * 300 * (27 * 64 + 576 * 64))
# a bandwidth kernel for contrast
xp = bf(B, 38, 38, 512).permute(0, 3, 1, 2)
g = torch.full((512,), 20.0, device="cuda")
try:
    under_load("pool4 + conv4_3_norm (HBM bound)", lambda: nat.pool2_l2_normalize(xp, g))
except Exception as exc:                                  # noqa: BLE001
    print(json.dumps({"pool2_l2_normalize": repr(exc)[:200]}))
