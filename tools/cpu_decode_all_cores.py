"""The all-core leg of bench.py's cpu_baseline ("also report an all-core figure by sharding batch items
across a process pool"): the NumPy port of the reference's decode_detections, one batch item per task, on every core this process
may run on.  Runs as its own process -- bench.py holds an initialised HIP runtime, which must not be forked --, reads
DIR/y.npy + DIR/kw.json, prints one JSON line.

    python tools/cpu_decode_all_cores.py DIR
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_KW = {}


def _start(y_small):
    """Pool initializer: every worker imports the oracle and decodes a few rows once, outside the timed region."""
    try:
        _decode_one(y_small)
    except Exception:                                   # noqa: BLE001 -- a worker dying here would make the pool respawn it forever;
        pass                                            # the same error surfaces from the first real task instead


def _decode_one(y):
    from oracle import np_oracle as orc
    with np.errstate(all="ignore"):
        return orc.decode_detections(y[None], **_KW)[0].shape[0]


def main(d):
    y = np.load(os.path.join(d, "y.npy"))
    _KW.update(json.load(open(os.path.join(d, "kw.json"))))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    n_proc = max(1, min(cores, len(y)))
    with mp.get_context("fork").Pool(n_proc, initializer=_start, initargs=(y[0][:16],)) as pool:   # fork: workers inherit _KW
        pool.map(_decode_one, [y[0][:16]] * n_proc)                 # returns once the workers are up: untimed
        t = time.perf_counter()
        rows = pool.map(_decode_one, list(y), chunksize=1)
        wall = time.perf_counter() - t
    print(json.dumps({"value": round(len(y) / wall, 4), "unit": "images/sec (decode_detections only)", "cores": n_proc,
                      "ms_per_img": round(1e3 * wall / len(y), 3), "images": int(len(y)), "detections": int(sum(rows)),
                      "note": "same port, one batch item per task on a %d-process pool" % n_proc}))


if __name__ == "__main__":
    main(sys.argv[1])
