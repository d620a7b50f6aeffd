cd $GRAFT_REPO_ROOT
SSD_TRAIN_GRAPH=1 SSD_TRAIN_TRACE=1 SSD_TRAIN_RAW=0 timeout 600 python - <<'P' 2>&1 | grep TRACE
import json, torch, bench_extra as bx
r = bx.train_leg(torch.device("cuda:0"), 0, 1, 32, steps=6, warmup=3, tame=True)
print("TRACE bench leg", json.dumps([v for _, v in r.get("loss_trace")]), r.get("ms_per_step"), r.get("launch"), r.get("error"))
P
SSD_TRAIN_GRAPH=1 SSD_TRAIN_RAW=0 timeout 600 python - <<'P' 2>&1 | grep TIMED
import json, torch, bench_extra as bx
r = bx.train_leg(torch.device("cuda:0"), 0, 1, 32, steps=6, warmup=3, tame=True)
print("TIMED bench leg", r.get("ms_per_step"), r.get("eager_ms_per_step"), r.get("first_loss"), r.get("final_loss"), r.get("error"))
P
