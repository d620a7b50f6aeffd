"""The data gradients of conv2_2 / conv3_2 (= conv3_3) / conv4_2 (= conv4_3) at batch 32 through the slab kernel: plain, masked (MSK
epilogue), masked + channel sums; and the passes they replace (relu_bwd_bias over both maps, channel sums over the masked map).  GPU box."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402


def timed(fn, reps=40):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return a.elapsed_time(e) / reps * 1e3


for name, (B, Cy, Cx, H) in (("conv2_2 -> conv2_1", (32, 128, 128, 150)), ("conv3_2 -> conv3_1", (32, 256, 256, 75)), ("conv4_2 -> conv4_1", (32, 512, 512, 38))):
    g = torch.Generator(device="cuda").manual_seed(H)
    gy = torch.randn((B, Cy, H, H), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((Cx, Cy, 3, 3), device="cuda", generator=g) * (2.0 / (9 * Cy)) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    act = torch.randn((B, Cx, H, H), device="cuda", generator=g).clamp_min(0).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    plain = nat.conv2d_same(gy, wt, None, dilation=1, relu=False, variant=7)
    masked = nat.conv3x3_halo_masked(gy, wt, act)
    for rep in range(3):
        row = {
            "plain": timed(lambda: nat.conv2d_same(gy, wt, None, dilation=1, relu=False, variant=7)),
            "masked": timed(lambda: nat.conv3x3_halo_masked(gy, wt, act)),
            "masked+sums": timed(lambda: nat.conv3x3_halo_masked(gy, wt, act, sums=True)),
            "relu_bwd_bias pass": timed(lambda: nat.relu_bwd_bias(plain, act, reduce=False)),
            "channel sums pass": timed(lambda: nat.channel_sums_partial(masked)),
        }
        print(name, "  ".join("%s %.1f us" % kv for kv in row.items()), flush=True)
