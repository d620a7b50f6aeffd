#!/bin/bash
# ONE parameterised GPU-box visit script (round 4: replaces the forty-odd tools/gpu_r03*.sh one-offs).
#
#   gpurun --timeout 900 -- 'bash tools/gpu.sh TAG step [step ...]'          everything lands under gpurun_out/TAG/
#
# steps (run in the order given; each is bounded by its own `timeout`):
#   tests                 the whole `-m gpu` suite              -> pytest_gpu.txt
#   tests=PATH[,PATH...]  only these test files / node ids      -> pytest_gpu.txt
#   bench                 the full bench line                   -> bench.json
#   benchq                bench.py --no-cpu-baseline --no-extra -> benchq.json   (the headline step only: ~40 s)
#   timeline[@LIB]        rocprofv3 kernel trace of 10 timed steps -> step_timeline[@LIB].json, bench_kernel_stats[@LIB].csv;
#                         @LIB runs the same code on tools/libssdhip_LIB.so (a within-visit A/B against an older build)
#   pmc_decode            FETCH_SIZE / WRITE_SIZE passes of the decode kernels -> decode_pmc_traffic.json
#   pmc_mfma              MFMA-busy counters per kernel of one eager forward   -> pmc_mfma_per_kernel.txt
#   train_timeline        one training step of bench_extra.train_leg by kernel (rocprofv3 trace) -> train_step_timeline.json
#   stats=SCRIPT          rocprofv3 --kernel-trace --stats around `python tools/SCRIPT.py` -> SCRIPT_kernel_stats.csv + SCRIPT.log
#   py=SCRIPT[:ARGS]      python tools/SCRIPT.py ARGS           -> SCRIPT.log
#   env:K=V               export K=V for the steps that follow (env:K= unsets)
set -u
TAG=${1:?tag}
shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for step in "$@"; do
  echo "=== $step"
  case "$step" in
    env:*) kv=${step#env:}; k=${kv%%=*}; v=${kv#*=}; if [ -z "$v" ]; then unset "$k"; else export "$k=$v"; fi ;;
    tests) timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.txt" 2>&1; tail -n 15 "$OUT/pytest_gpu.txt" ;;
    tests=*) timeout 1500 python -m pytest $(echo "${step#tests=}" | tr ',' ' ') -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.txt" 2>&1; tail -n 15 "$OUT/pytest_gpu.txt" ;;
    bench) timeout 1200 python bench.py > "$OUT/bench.json" 2> "$OUT/bench_err.log"; tail -c 3000 "$OUT/bench.json"; tail -n 3 "$OUT/bench_err.log" ;;
    benchq) timeout 600 python bench.py --no-cpu-baseline --no-extra > "$OUT/benchq.json" 2> "$OUT/benchq_err.log"; head -c 600 "$OUT/benchq.json"; echo; tail -n 3 "$OUT/benchq_err.log" ;;
    timeline*)
      sfx=""; lib=""
      if [ "$step" != "timeline" ]; then lib=${step#timeline@}; sfx="@$lib"; fi
      ( cd /tmp
        [ -n "$lib" ] && export SSDHIP_LIB=$R/tools/libssdhip_$lib.so
        timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace$sfx" -o bench -- \
          python "$R/bench.py" --steps 10 --warmup 5 --no-cpu-baseline --no-extra > "$OUT/trace_bench$sfx.json" 2> "$OUT/trace_err$sfx.log" )
      f=$(find "$OUT/trace$sfx" -name "*kernel_trace.csv" | head -1)
      python tools/step_timeline.py "$f" "$OUT/step_timeline$sfx.json"
      cp "$(find "$OUT/trace$sfx" -name "*kernel_stats.csv" | head -1)" "$OUT/bench_kernel_stats$sfx.csv" 2>/dev/null
      head -c 300 "$OUT/trace_bench$sfx.json"; echo ;;
    pmc_decode)
      for C in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp; REPS=5 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o dec -- python "$R/tools/pmc_decode.py" > "$OUT/pmc_$C.log" 2>&1 )
      done
      python tools/pmc_summary.py "$OUT/decode_pmc_traffic.json" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" > "$OUT/pmc_summary.log" 2>&1; tail -n 5 "$OUT/pmc_summary.log" ;;
    pmc_mfma)
      ( cd /tmp; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 \
          --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o fwd -- python "$R/bench.py" --graph 0 --steps 3 --warmup 8 --no-cpu-baseline --no-extra > "$OUT/pmc_mfma.log" 2>&1 )
      python tools/pmc_fold.py "$OUT/pmc_mfma" ssdhip > "$OUT/pmc_mfma_per_kernel.txt" 2>&1; head -n 40 "$OUT/pmc_mfma_per_kernel.txt" ;;
    train_timeline)
      ( cd /tmp; SSD_TRAIN_RAW=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_train" -o train -- python "$R/tools/time_train_leg.py" > "$OUT/train_leg.log" 2>&1 )
      python tools/train_timeline.py "$(find "$OUT/trace_train" -name "*kernel_trace.csv" | head -1)" "$OUT/train_step_timeline.json" rowmax_kernel | head -45
      tail -n 2 "$OUT/train_leg.log" ;;
    stats=*)
      s=${step#stats=}
      ( cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$s" -o "$s" -- python "$R/tools/$s.py" > "$OUT/$s.log" 2>&1 )
      cp "$(find "$OUT/trace_$s" -name "*kernel_stats.csv" | head -1)" "$OUT/${s}_kernel_stats.csv" 2>/dev/null
      tail -n 12 "$OUT/$s.log"; head -n 12 "$OUT/${s}_kernel_stats.csv" ;;
    py=*)
      sa=${step#py=}; s=${sa%%:*}; a=""; [ "$sa" != "$s" ] && a=${sa#*:}
      timeout 900 python "tools/$s.py" $(echo "$a" | tr ',' ' ') > "$OUT/$s.log" 2>&1; tail -n 40 "$OUT/$s.log" ;;
    *) echo "unknown step $step" ;;
  esac
done
find "$OUT" -name "*.csv" -size +5M -delete
find "$OUT" -name "*.db" -delete
exit 0
