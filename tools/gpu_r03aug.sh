cd $GRAFT_REPO_ROOT
timeout 200 python - <<'P' 2>&1 | tail -2
import json, torch, bench_extra as bx
print(json.dumps(bx.augmentation_leg(torch.device("cuda:0"), 32, True)))
P
