"""Helpers shared by the oracle-vs-golden and HIP-vs-oracle tests."""
import ast
import os

import numpy as np

from ssd_keras_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def unragged(cat, off):
    return [cat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def kw_of(z, name):
    return ast.literal_eval(str(z[name + "_kw"]))


def sort_rows(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[0] == 0:
        return a.reshape(0, a.shape[-1] if a.ndim == 2 else 0)
    order = np.lexsort(tuple(a[:, c] for c in range(a.shape[1] - 1, -1, -1)))
    return a[order]


def dets_equal(got, want, exact=True, rtol=1e-4, atol=1e-4, n_meta=2):
    """Two detection lists (one array per image) hold the same rows, order ignored.
    Leading `n_meta` columns (ids, class, conf) must match exactly; box columns exactly
    or within tolerance."""
    assert len(got) == len(want), (len(got), len(want))
    for b, (g, w) in enumerate(zip(got, want)):
        g, w = sort_rows(g), sort_rows(w)
        assert g.shape[0] == w.shape[0], "image %d: %d rows vs %d" % (b, g.shape[0], w.shape[0])
        if g.shape[0] == 0:
            continue
        assert g.shape == w.shape, (g.shape, w.shape)
        if exact:
            assert np.array_equal(g, w), "image %d differs: max abs %g" % (b, np.abs(g - w).max())
        else:
            assert np.array_equal(g[:, :n_meta], w[:, :n_meta]), "image %d: ids/classes/conf differ" % b
            np.testing.assert_allclose(g[:, n_meta:], w[:, n_meta:], rtol=rtol, atol=atol)


def local_exp_matches_golden():
    z = load("decoder")
    return np.array_equal(np.exp(z["exp_probe_in"]), z["exp_probe_out"])


CFGS = dict(tiny=syn.TINY, ssd7=syn.SSD7_300, ssd300=syn.SSD300_VOC, ssd512=syn.SSD512_COCO)
