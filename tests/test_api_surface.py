"""the package mirrors the reference's Python call surface.  tests/golden/api_surface.json holds parameter names,
order and defaults of every mirrored callable, read from the reference's source with `ast` (tests/golden/make_golden.py api_surface;
the TensorFlow / Keras modules need not import for that).  The package's own source must carry every reference parameter, in the
reference's position, with the reference's default; it may ADD defaulted parameters behind them (device-side conveniences)."""
import json
import os

from tests import api_surface

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "ssd_keras_amd")

# callables the package inherits from a class of ANOTHER module (ast does not follow imports): the reference's signature must then be
# the same as that of the class it is inherited from
INHERITED = {("keras_layers/keras_layer_DecodeDetectionsFast.py", "DecodeDetectionsFast.__init__"):
             ("keras_layers/keras_layer_DecodeDetections.py", "DecodeDetections.__init__")}


def _plain(params):
    return [p for p in params if not p[0].startswith("*")]


def test_every_reference_parameter_is_there_in_place_with_its_default():
    want = json.load(open(os.path.join(HERE, "golden", "api_surface.json")))
    got = api_surface.extract(PKG)
    assert set(want) == set(api_surface.SURFACE), "fixture is stale: rerun tests/golden/make_golden.py api_surface"
    problems = []
    for mod, names in want.items():
        for q, ref in names.items():
            ref = [tuple(p) for p in ref]
            ours = got[mod].get(q)
            if ours is None and (mod, q) in INHERITED:
                base_mod, base_q = INHERITED[(mod, q)]
                assert [tuple(p) for p in want[base_mod][base_q]] == ref, (mod, q, "differs from the class it is inherited from")
                ours = got[base_mod][base_q]
            if ours is None:
                problems.append((mod, q, "missing"))
                continue
            rp, op = _plain(ref), _plain(ours)
            if [p[0] for p in op[:len(rp)]] != [p[0] for p in rp]:
                problems.append((mod, q, "parameters", [p[0] for p in rp], [p[0] for p in op]))
                continue
            for (name, default), (_, mine) in zip(rp, op):
                if default != mine:
                    problems.append((mod, q, name, "default", default, mine))
            for extra in op[len(rp):]:
                if extra[1] is None:
                    problems.append((mod, q, extra[0], "an added parameter needs a default"))
            for star in (p[0] for p in ref if p[0].startswith("*")):
                if star[:2] == "**" and not any(p[0].startswith("**") for p in ours):
                    problems.append((mod, q, star, "keyword catch-all missing"))
    assert not problems, problems


def test_loss_helper_methods_follow_their_definitions():
    """SSDLoss.smooth_L1_loss / log_loss (reference keras_ssd_loss.py:53-96) against the oracle's NumPy restatement."""
    import numpy as np
    import torch
    from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss
    rng = np.random.RandomState(0)
    t, p = rng.randn(3, 50, 4).astype(np.float32) * 2, rng.randn(3, 50, 4).astype(np.float32) * 2
    d = np.abs(t - p)
    want = np.where(d < 1.0, 0.5 * (t - p) ** 2, d - 0.5).sum(-1)
    got = SSDLoss().smooth_L1_loss(torch.from_numpy(t), torch.from_numpy(p)).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    yt = np.eye(7, dtype=np.float32)[rng.randint(0, 7, size=(3, 50))]
    yp = rng.uniform(0, 1, size=(3, 50, 7)).astype(np.float32)
    yp[0, 0] = 0.0                                               # the 1e-15 floor
    want = -(yt * np.log(np.maximum(yp, 1e-15))).sum(-1)
    got = SSDLoss().log_loss(torch.from_numpy(yt), torch.from_numpy(yp)).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    x = torch.from_numpy(p).requires_grad_(True)
    SSDLoss().smooth_L1_loss(torch.from_numpy(t), x).sum().backward()
    assert x.grad is not None and bool(torch.isfinite(x.grad).all())
