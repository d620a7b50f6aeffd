"""
ORACLE -- TEST INFRASTRUCTURE ONLY (see np_oracle.py).  Runs an oracle decoder over the items of a batch on a process pool.

The oracle's greedy NMS is a Python loop (1-2 s per dense SSD300 image); the decoders are independent per batch item
(ssd_output_decoder.py:205), so a full batch-32 comparison costs a few seconds of wall time on the GPU box's host cores.  The
pool lives in its OWN interpreter (`python -m oracle.pool DIR`): the caller holds an initialised HIP runtime, which must not be
forked.  This is for CHECKING at full batch size; timed CPU baselines call np_oracle directly on one core.

    from oracle import pool
    rows = pool.decode("decode_detections", y_pred, dict(confidence_thresh=0.01, ...))     # list of per-image arrays
    out  = pool.decode("decode_detections_layer", y_pred, kw)                               # (B, top_k, 6) float32
"""
from __future__ import annotations

import json
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUNCS = ("decode_detections", "decode_detections_fast", "decode_detections_layer")


def decode(fn_name, y_pred, kw, procs=None, timeout_s=900):
    """`np_oracle.<fn_name>(y_pred, **kw)` with one batch item per task.  Returns what the function returns for the whole batch."""
    if fn_name not in FUNCS:
        raise ValueError("unknown oracle function %r" % fn_name)
    y_pred = np.ascontiguousarray(y_pred)
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "y.npy"), y_pred)
        with open(os.path.join(d, "spec.json"), "w") as f:
            json.dump({"fn": fn_name, "kw": kw, "procs": procs}, f)
        run = subprocess.run([sys.executable, "-m", "oracle.pool", d], cwd=ROOT, capture_output=True, text=True, timeout=timeout_s)
        if run.returncode != 0:
            raise RuntimeError("oracle pool failed: %s" % run.stderr.strip()[-500:])
        with open(os.path.join(d, "out.pkl"), "rb") as f:
            res = pickle.load(f)
    if fn_name == "decode_detections_layer":
        return np.concatenate(res, axis=0)
    return [r[0] for r in res]


_SPEC = {}


def _one(y):
    from oracle import np_oracle as orc
    with np.errstate(all="ignore"):
        return getattr(orc, _SPEC["fn"])(y[None], **_SPEC["kw"])


def _main(d):
    import multiprocessing as mp
    y = np.load(os.path.join(d, "y.npy"))
    _SPEC.update(json.load(open(os.path.join(d, "spec.json"))))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    n_proc = max(1, min(_SPEC.get("procs") or cores, len(y)))
    if n_proc == 1:
        res = [_one(item) for item in y]
    else:
        with mp.get_context("fork").Pool(n_proc) as pool:            # fork: the workers inherit _SPEC
            res = pool.map(_one, list(y), chunksize=1)
    with open(os.path.join(d, "out.pkl"), "wb") as f:
        pickle.dump(res, f)


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    _main(sys.argv[1])
