#!/bin/bash
# Round-3 visit P: the reference-precision path on the slab kernel: tests + the fp32x3 leg.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03v
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_precise_gpu.py tests/test_conv_gpu.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest.txt 2>&1
grep -E "float16 x 3|passed|failed|FAILED" $OUT/pytest.txt | tail -n 12 | cut -c1-220
python - <<'P'
import json, torch, bench_extra as bx
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
a = bx.fp32_forward_leg(dev, 32)
b = bx.fp32x3_forward_leg(dev, 32, a)
print(json.dumps({"conv_roofline_fp32": {k: v for k, v in a.items() if k != "note"}, "conv_roofline_fp32x3": {k: v for k, v in b.items() if k not in ("note", "dtype")}}))
json.dump({"conv_roofline_fp32": a, "conv_roofline_fp32x3": b}, open("gpurun_out/r03v/fp32x3_leg.json", "w"), indent=1)
P
