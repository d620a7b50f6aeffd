#!/bin/bash
# Round-3 visit X2: the fused pooling + ReLU backward: tests, then the training leg with and without it, alternating.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03zj
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_train_glue_gpu.py tests/test_end_to_end_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_train.txt 2>&1
tail -n 3 $OUT/pytest_train.txt
for rep in 1 2; do
  for f in 1 0; do
    SSDHIP_NO_FUSED_POOL_BWD=$f SSD_TRAIN_RAW=0 timeout 600 python - $f <<'P'
import json, sys, torch, bench_extra as bx
r = bx.train_leg(torch.device("cuda:0"), 0, 1, 32, steps=6, warmup=3, tame=True)
print("NO_FUSED_POOL_BWD", sys.argv[1], json.dumps({k: r.get(k) for k in ("ms_per_step", "eager_ms_per_step", "first_loss", "final_loss", "launch", "error")}))
P
  done
done 2>&1 | grep NO_FUSED
