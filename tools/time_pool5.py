"""pool5 (3x3 / 1 'same' max pooling of the 19 x 19 x 512 map, batch 32): event timing of back-to-back launches.  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

for shape in ((32, 512, 19, 19), (16, 512, 32, 32)):
    x = torch.randn(shape, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        nat.bias_act_maxpool(x, None, 3, 1, 1, False, relu=False)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        nat.bias_act_maxpool(x, None, 3, 1, 1, False, relu=False)
    e.record()
    e.synchronize()
    print(shape, "%.1f us per launch" % (a.elapsed_time(e) / 50 * 1e3), flush=True)
