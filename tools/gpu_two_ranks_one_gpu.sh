#!/bin/bash
# The N > 1 code path of bench.py (launcher environment, process group, barriers, max over ranks, the DDP training leg with libssdhip's
# autograd functions) exercised on a ONE-GPU box: two ranks share GPU 0 and talk over gloo.  Not a measurement -- RCCL itself and a
# second GPU are what this cannot cover.   bash tools/gpu_two_ranks_one_gpu.sh [OUTDIR]
set -u
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/two_ranks}
mkdir -p $OUT
cd $R
SSD_BENCH_BACKEND=gloo SSD_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_two_ranks.json 2> $OUT/bench_two_ranks.err
echo "rc=$?"
tail -c 600 $OUT/bench_two_ranks.err
python - $OUT/bench_two_ranks.json <<'P'
import json, sys
lines = [l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")]
print("json lines:", len(lines))
d = json.loads(lines[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling")})
t = d.get("train_step")
print("train_step", json.dumps({k: t.get(k) for k in ("n_gpus", "ms_per_step", "eager_ms_per_step", "rank_step_ms_min_max", "allreduce_buckets", "allreduce_bytes_per_step", "first_loss", "final_loss", "launch", "error")}) if isinstance(t, dict) else t)
P
