#!/bin/bash
# Round-3 visit M: head row builder without divisions, wide-row scan kernel: tests + timings.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_decode_layer_gpu.py tests/test_decode_fullsize_gpu.py tests/test_layers_gpu.py tests/test_end_to_end_gpu.py tests/test_precise_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest.txt 2>&1
tail -n 6 $OUT/pytest.txt
for lib in tools/libssdhip_r02.so ssd_keras_amd/libssdhip.so; do
  echo "== $lib"
  SSDHIP_LIB=$R/$lib MODEL=0 timeout 300 python tools/time_decode.py 2>&1 | grep "ssd512" | cut -c1-260
done
python - <<'P'
import json, torch, bench_extra as bx
dev = torch.device("cuda:0")
r = bx.ssd512_decode_leg(dev, with_cpu=False)
print(json.dumps(r)[:1500])
P
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_quick.json 2> $OUT/bench_err.log
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03s/bench_quick.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], "decode_in_step", d["roofline"]["decode_ms_in_step"])
P
