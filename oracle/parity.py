"""
ORACLE -- TEST INFRASTRUCTURE ONLY (see np_oracle.py).  Comparators for decoder outputs.

`decode_detections` (ssd_encoder_decoder/ssd_output_decoder.py:217-221) cuts an image's NMS survivors to `top_k` rows with
`np.argpartition`: when several survivors share the k-th confidence, WHICH of them make the cut is arbitrary in the reference
itself (it depends on NumPy's introselect), so a row-for-row comparison of two correct implementations fails exactly there.
`topk_tie_aware` compares what the reference does define:
  * every row strictly above the k-th confidence must be present on both sides, bit for bit;
  * at the k-th confidence only the COUNT is defined, and every row the implementation kept there must be a member of the
    uncut (`top_k='all'`) survivor set -- the oracle's and, when given, the implementation's own.
The Keras layers (keras_layer_DecodeDetections.py:238-251, `tf.nn.top_k(sorted=True)`) break ties deterministically (lower
position in the class-major padded array first), so their outputs are compared with `np.array_equal`.
"""
from __future__ import annotations

import numpy as np


def canon(a, width=6):
    """Rows sorted lexicographically (order-free comparison); empty containers -> (0, width)."""
    a = np.array(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[0] == 0:
        return np.zeros((0, width))
    a = a + 0.0                                          # -0.0 -> +0.0
    a[np.isnan(a)] = np.nan                              # one NaN bit pattern (overflowed boxes decode to inf - inf)
    return a[np.lexsort(tuple(a[:, c] for c in range(a.shape[1] - 1, -1, -1)))]


def _same(a, b):
    return a.shape == b.shape and bool(np.array_equal(a, b, equal_nan=True))


def _row_set(a):
    return {r.tobytes() for r in np.ascontiguousarray(canon(a, 6))}


def topk_tie_aware(got, ref_all, top_k, got_all=None, conf_col=1):
    """One image.  got: rows the implementation returned for `top_k`; ref_all: the oracle's rows with top_k='all';
    got_all: the implementation's rows with top_k='all' (optional).  Returns a dict of findings; `ok` is their conjunction."""
    width = 6 if np.asarray(ref_all).ndim != 2 else np.asarray(ref_all).shape[1]
    g, r = canon(got, width), canon(ref_all, width)
    res = {"survivors": int(r.shape[0]), "rows": int(g.shape[0]), "tie_group_size": 0, "rows_at_cut": 0}
    if got_all is not None:
        res["uncut_survivors_equal"] = _same(canon(got_all, width), r)
    if r.shape[0] <= top_k:                              # nothing to cut: plain set equality
        same = _same(g, r)
        res.update(rows_above_cut_equal=same, count_ok=g.shape[0] == r.shape[0], tie_members_valid=same)
    else:
        kth = np.sort(r[:, conf_col])[-top_k]            # the top_k-th largest confidence among all survivors
        g_above, r_above = g[g[:, conf_col] > kth], r[r[:, conf_col] > kth]
        g_tie, r_tie = g[g[:, conf_col] == kth], r[r[:, conf_col] == kth]
        members = _row_set(r_tie)
        if got_all is not None:
            members &= _row_set(canon(got_all, width))
        tie_rows = [row.tobytes() for row in np.ascontiguousarray(g_tie)]
        res.update(rows_above_cut_equal=_same(g_above, r_above),
                   count_ok=bool(g.shape[0] == top_k and g_tie.shape[0] == top_k - r_above.shape[0]),
                   tie_members_valid=bool(all(t in members for t in tie_rows) and len(set(tie_rows)) == len(tie_rows)),
                   tie_group_size=int(r_tie.shape[0]), rows_at_cut=int(g_tie.shape[0]))
    res["ok"] = bool(res["rows_above_cut_equal"] and res["count_ok"] and res["tie_members_valid"]
                     and res.get("uncut_survivors_equal", True))
    return res


def decode_parity(got_list, ref_all_list, top_k, got_all_list=None):
    """A batch: aggregates `topk_tie_aware` over the images into the flags bench.py prints."""
    per = [topk_tie_aware(g, r, top_k, None if got_all_list is None else got_all_list[i])
           for i, (g, r) in enumerate(zip(got_list, ref_all_list))]
    out = {"images": len(per),
           "rows_above_cut_equal": all(p["rows_above_cut_equal"] for p in per),
           "count_ok": all(p["count_ok"] for p in per),
           "tie_members_valid": all(p["tie_members_valid"] for p in per),
           "tie_group_size": [p["tie_group_size"] for p in per],
           "rows_at_cut": [p["rows_at_cut"] for p in per],
           "survivors": [p["survivors"] for p in per]}
    if got_all_list is not None:
        out["uncut_survivors_equal"] = all(p["uncut_survivors_equal"] for p in per)
    out["ok"] = all(p["ok"] for p in per)
    return out


def layer_parity(got, want):
    """(B, top_k, 6) float32 outputs of the DecodeDetections layer semantics: deterministic ties -> exact comparison
    (NaN-aware: an overflowed box decodes to NaN coordinates on both sides)."""
    got, want = np.asarray(got), np.asarray(want)
    same = got.shape == want.shape and bool(np.array_equal(got, want, equal_nan=True))
    bad = []
    if not same and got.shape == want.shape:
        bad = [int(b) for b in range(got.shape[0]) if not np.array_equal(got[b], want[b], equal_nan=True)]
    return {"images": int(got.shape[0]), "equal": same, "images_differing": bad,
            "rows": int((want[:, :, 0] != 0).sum())}
