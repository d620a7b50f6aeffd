import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import np_oracle as orc
from tests import util
from tests.test_oracle_golden import _y_pred_for
from ssd_keras_amd.ssd_encoder_decoder import ssd_output_decoder as d
z = util.load("decoder")
for name in [str(s) for s in z["cases"]]:
    if str(z[name + "_fn"]) != "decode_detections":
        continue
    kw = util.kw_of(z, name)
    y = _y_pred_for(z, name)
    if y.dtype != np.float32:
        continue
    got = d.decode_detections(y, **kw)
    want = orc.decode_detections(y, exp_mode="det", **kw)
    for b, (g, w) in enumerate(zip(got, want)):
        if g.shape != w.shape or not np.array_equal(util.sort_rows(g), util.sort_rows(w)):
            print("MISMATCH", name, "img", b, g.shape, w.shape, kw)
            if g.size and w.size:
                for c in range(1, y.shape[2] - 12):
                    gc, wc = g[g[:, 0] == c], w[w[:, 0] == c]
                    if gc.shape != wc.shape:
                        print("  class", c, "got", gc.shape[0], "want", wc.shape[0])
                        ws = {tuple(r) for r in wc.tolist()}
                        extra = [r for r in gc.tolist() if tuple(r) not in ws]
                        print("   first extras:", extra[:3])
                        if extra:
                            e = np.array(extra[0])
                            ious = orc.iou(wc[:, 2:], e[2:], "corners", "element-wise", kw.get("border_pixels", "half"))
                            hi = wc[:, 1] > e[1]
                            print("   max IoU with higher-scored kept:", ious[hi].max() if hi.any() else None, "n higher", hi.sum(), "pos in got", int(np.nonzero((gc == e).all(1))[0][0]))
            break
    else:
        continue
    break
else:
    print("all decode_detections golden cases match")

# deeper: class-4 kept order with anchor ids
import torch
from ssd_keras_amd import _native as nat
name = "tiny_centroids_float32_b0_all"
kw = util.kw_of(z, name)
y = _y_pred_for(z, name)[:1]
out, count, aidx = nat.decode(torch.from_numpy(y).cuda(), 0.05, 0.3, 0, 0, False, nat.SEM_NUMPY, "centroids", False, None, None, "half", nat.F64, 5 * 340, want_anchor_idx=True)
g = np.concatenate([aidx[0, :int(count[0])].cpu().numpy()[:, None].astype(float), out[0, :int(count[0])].cpu().numpy()], axis=1)
w = orc.decode_detections(y, exp_mode="det", with_anchor_index=True, **kw)[0]
for c in (4,):
    gc, wc = g[g[:, 1] == c], w[w[:, 1] == c]
    n = int((y[0, :, c] > 0.05).sum())
    print("class", c, "n cand", n, "got", gc.shape[0], "want", wc.shape[0])
    k = 0
    while k < min(len(gc), len(wc)) and gc[k, 0] == wc[k, 0]:
        k += 1
    print(" first divergence at kept position", k, "gpu id/score", gc[k, [0, 2]], "oracle id/score", wc[k, [0, 2]])
    order = np.argsort(-y[0, :, c].astype(np.float64), kind="stable")
    order = order[y[0, order, c] > 0.05]
    rank = {int(a): i for i, a in enumerate(order)}
    print(" sorted-rank of gpu kept[%d..%d]:" % (max(0, k - 3), k + 5), [rank[int(a)] for a in gc[max(0, k - 3):k + 6, 0]])
    print(" sorted-rank of oracle kept:", [rank[int(a)] for a in wc[max(0, k - 3):k + 6, 0]])
