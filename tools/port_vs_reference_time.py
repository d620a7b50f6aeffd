"""BUILD CONTAINER ONLY (the reference is mounted at /root/reference here and exists nowhere else): wall time of the oracle's port of
decode_detections against the REAL reference's on the same arrays -- the ratio bench.py's cpu_baseline cannot measure on the GPU box
(VERDICT r5 weak 11).  Writes profiles/<tag>_port_vs_reference_time_ratio.json, which bench.py replays (labelled as such).
    python tools/port_vs_reference_time.py r06"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("SSD_REFERENCE", "/root/reference"))
np.float, np.int, np.bool = float, int, bool          # noqa: aliases the reference still uses

from ssd_encoder_decoder.ssd_output_decoder import decode_detections as ref_decode      # noqa: E402

from oracle import np_oracle as orc                   # noqa: E402
from ssd_keras_amd import synthetic as syn            # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
enc = orc.EncoderOracle(**syn.SSD300_VOC)
anchors = enc.generate_encoding_template(1)[0, :, -8:]
out = {"where": "build container, %d host cpus, one core used" % os.cpu_count(), "cases": {}}
kw = dict(confidence_thresh=0.01, iou_threshold=0.45, top_k=200, normalize_coords=True, img_height=300, img_width=300)
for name, bias, n_img in (("sparse_bias7", 7.0, 6), ("mid_bias3", 3.0, 2), ("dense_bias0", 0.0, 1)):
    y = syn.make_y_pred(anchors, n_img, enc.n_classes, bias=bias, seed=1234)
    with np.errstate(all="ignore"):
        ref_decode(y[:1], **kw); orc.decode_detections(y[:1], **kw)                     # warm
        t = time.perf_counter(); a = ref_decode(y, **kw); t_ref = time.perf_counter() - t
        t = time.perf_counter(); b = orc.decode_detections(y, **kw); t_port = time.perf_counter() - t
    same = all(np.array_equal(np.sort(u, axis=0), np.sort(v, axis=0)) if (u.size and v.size) else u.size == v.size for u, v in zip(a, b))
    out["cases"][name] = {"images": n_img, "reference_s": round(t_ref, 3), "port_s": round(t_port, 3), "port_over_reference": round(t_port / t_ref, 3),
                          "identical_rows": bool(same)}
out["port_over_reference_range"] = [min(c["port_over_reference"] for c in out["cases"].values()), max(c["port_over_reference"] for c in out["cases"].values())]
path = os.path.join(ROOT, "profiles", "%s_port_vs_reference_time_ratio.json" % tag)
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
