cd $GRAFT_REPO_ROOT
for g in 1 0; do
SSD_TRAIN_GRAPH=$g SSD_TRAIN_TRACE=1 SSD_TRAIN_RAW=0 timeout 600 python - $g <<'P' 2>&1 | grep TRACE
import json, sys, torch, bench_extra as bx
r = bx.train_leg(torch.device("cuda:0"), 0, 1, 32, steps=6, warmup=3, tame=True)
print("TRACE graph=%s" % sys.argv[1], json.dumps(r.get("loss_trace")), r.get("error"))
P
done
