#!/bin/bash
# Round-3 visit R: reference-precision leg after the first-layer kernel; head tile sweep of the fused scan; whole GPU suite.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03y
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt
python - <<'P'
import json, torch, bench_extra as bx
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
a = bx.fp32_forward_leg(dev, 32)
b = bx.fp32x3_forward_leg(dev, 32, a)
print(json.dumps({"fp32": {k: v for k, v in a.items() if k != "note"}, "fp32x3": {k: v for k, v in b.items() if k not in ("note", "dtype")}}))
json.dump({"conv_roofline_fp32": a, "conv_roofline_fp32x3": b}, open("gpurun_out/r03y/fp32x3_leg.json", "w"), indent=1)
P
for kb in 60 24 12; do
  SSDHIP_LIB=tools/libssdhip_prof.so SSDHIP_HEADS_LDS_KB=$kb timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_heads_lds$kb.json 2> $OUT/bench_err.log
  python - $OUT/bench_heads_lds$kb.json $kb <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("heads tile LDS KB", sys.argv[2], "step", d["ms_per_step"], "decode_in_step", d["roofline"]["decode_ms_in_step"])
P
done
