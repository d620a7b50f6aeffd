"""fc7 / conv6_1 / conv6_2 alone (GPU box): the image-resident kernel's general form against the implicit-GEMM kernels, events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)


def mk(B, H, W, Cin, Cout, k):
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / (k * k * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    b = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    return x, w, b


def ev(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return 1e3 * best


for name, (B, H, W, Cin, Cout, k, s, p) in {"fc7": (32, 19, 19, 1024, 1024, 1, 1, 0), "conv6_1": (32, 19, 19, 1024, 256, 1, 1, 0),
                                             "conv6_2": (32, 19, 19, 256, 512, 3, 2, 1), "fc6": (32, 19, 19, 512, 1024, 3, 1, 6)}.items():
    x, w, b = mk(B, H, W, Cin, Cout, k)
    d = 6 if name == "fc6" else 1
    t_img = ev(lambda: nat.conv2d_image(x, w, b, stride=s, padding=p, dilation=d, relu=True))
    t_ig = ev(lambda: nat.conv2d(x, w, b, stride=s, padding=p, dilation=d, relu=True, variant=5))
    print("%-8s image %.1f us   igemm5 %.1f us" % (name, t_img, t_ig), flush=True)
