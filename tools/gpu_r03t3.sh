cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for m in unset 3 0; do
  if [ $m = unset ]; then unset SSDHIP_HEAD_OVERLAP; else export SSDHIP_HEAD_OVERLAP=$m; fi
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HEAD_OVERLAP $m step_ms', d['ms_per_step'], 'decode_in_step', d['roofline']['decode_ms_in_step'])"
done; done
