#!/bin/bash
# Round-3 end visit: the whole GPU suite, the full bench line, rocprofv3 kernel stats + one step's timeline, decode PMC traffic,
# per-kernel MFMA-busy counters, encoder / loss kernel stats -- all on the commit that is pushed.  bash tools/gpu_final_r03.sh TAG
set -u
TAG=${1:-r03z}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
tail -n 3 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench_err.log
head -c 400 $OUT/bench.json; echo
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra > $OUT/trace_bench.json 2> $OUT/trace_err.log
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $f $OUT/step_timeline.json > /dev/null 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  REPS=5 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o dec -- python $R/tools/pmc_decode.py > $OUT/pmc_$C.log 2>&1
done
python $R/tools/pmc_summary.py $OUT/decode_pmc_traffic.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_summary.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    --kernel-trace --output-format csv -d $OUT/pmc_mfma -o fwd -- python $R/bench.py --graph 0 --steps 3 --warmup 8 --no-cpu-baseline --no-extra > $OUT/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_enc -o enc -- python $R/tools/time_encoder.py > $OUT/time_encoder.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_loss -o loss -- python $R/tools/time_loss.py > $OUT/time_loss.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_x3 -o x3 -- python $R/tools/prof_x3.py > $OUT/prof_x3.log 2>&1
cp $(find $OUT/trace_x3 -name "*kernel_stats.csv" | head -1) $OUT/x3_kernel_stats.csv 2>/dev/null
cd $R
python tools/pmc_fold.py $OUT/pmc_mfma ssdhip > $OUT/pmc_mfma_per_kernel.txt 2>&1
cp $(find $OUT/trace_enc -name "*kernel_stats.csv" | head -1) $OUT/encoder_kernel_stats.csv 2>/dev/null
cp $(find $OUT/trace_loss -name "*kernel_stats.csv" | head -1) $OUT/loss_kernel_stats.csv 2>/dev/null
find $OUT -name "*.csv" -size +5M -delete
find $OUT -name "*.db" -delete
grep case $OUT/time_encoder.log $OUT/time_loss.log
python - $OUT <<'P'
import json, sys, os
o = sys.argv[1]
d = json.loads(open(os.path.join(o, "bench.json")).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["kernel_ms"], "decode_in_step", d["roofline"]["decode_ms_in_step"])
print("conv", d["conv_roofline"]["forward_ms"], d["conv_roofline"]["frac"])
for k in ("encoder", "loss", "decode_sparse", "ssd512_decode", "conv_roofline_fp32", "conv_roofline_fp32x3", "train_step"):
    v = d.get(k)
    print(k, json.dumps(v)[:600] if v is not None else None)
P
head -40 $OUT/pmc_mfma_per_kernel.txt
