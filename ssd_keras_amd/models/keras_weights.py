"""Keras-layout weights <-> the PyTorch SSD modules.

The reference ships weights as Keras HDF5 files (`model.load_weights(path, by_name=True)`, ssd300_training.ipynb:120-140)
whose layers are named exactly like the layers in models/keras_ssd300.py:274-361 (`conv1_1` ... `conv9_2`, `fc6`, `fc7`,
`conv4_3_norm`, `<source>_mbox_conf`, `<source>_mbox_loc`; keras_ssd512.py adds `conv10_1/2`; keras_ssd7.py uses
`conv1..7`, `bn1..7`, `classes4..7`, `boxes4..7`).  Layouts differ: a Keras Conv2D kernel is HWIO `(kh, kw, in, out)`,
torch's is OIHW; BatchNormalization stores `gamma, beta, moving_mean, moving_variance`.

`keras_layer_map(model)` gives name -> list of (parameter/buffer, to_torch, to_keras) in Keras' `weight_names` order;
`load_keras_weights(model, source)` takes a `{layer_name: [arrays in Keras order]}` dict, the path of an `.npz` container
(keys "<layer>/<index>": `save_keras_weights_npz`, or `NPZ_CONVERSION` run once next to the Keras file) or the path of a Keras `.h5`
weight file -- read with h5py where it is installed, otherwise with this package's own reader of the HDF5 subset Keras writes
(models/hdf5_lite.py, round 6; `save_keras_weights_h5` writes the same container);
`export_keras_weights(model)` is the inverse (dict).  `by_name` semantics as in Keras: layers missing from the source keep
their initialisation, layers whose shapes do not match raise ValueError.
"""
from __future__ import annotations

import numpy as np
import torch


def _conv(c):
    items = [(c.weight, lambda a: np.transpose(a, (3, 2, 0, 1)), lambda t: np.transpose(t, (2, 3, 1, 0)))]
    if c.bias is not None:
        items.append((c.bias, lambda a: a, lambda t: t))
    return items


def _bn(b):
    ident = (lambda a: a, lambda t: t)
    return [(b.weight,) + ident, (b.bias,) + ident, (b.running_mean,) + ident, (b.running_var,) + ident]


def keras_layer_map(model):
    """Keras layer name -> [(tensor, keras->torch, torch->keras), ...] for every weighted layer of an SSD300 / SSD512 / SSD7."""
    from .keras_ssd7 import SSD7
    m = {}
    if isinstance(model, SSD7):
        for i in range(7):
            m["conv%d" % (i + 1)] = _conv(model.convs[i])
            m["bn%d" % (i + 1)] = _bn(model.bns[i])
        for i in range(4):
            m["classes%d" % (i + 4)] = _conv(model.conf_heads[i])
            m["boxes%d" % (i + 4)] = _conv(model.loc_heads[i])
        return m
    for name, mod in model.named_children():
        if isinstance(mod, torch.nn.Conv2d):
            m[name] = _conv(mod)                               # conv1_1 ... conv10_2, fc6, fc7: attribute names == Keras names
    m["conv4_3_norm"] = [(model.conv4_3_norm.gamma, lambda a: a, lambda t: t)]
    for i, src in enumerate(model.NAMES):
        m[src + "_mbox_conf"] = _conv(model.conf_heads[i])
        m[src + "_mbox_loc"] = _conv(model.loc_heads[i])
    return m


def _strings(values):
    return [v.decode("utf-8") if isinstance(v, bytes) else str(v) for v in np.asarray(values).reshape(-1).tolist()]


def _chunked_attr(attrs, name):
    """Keras splits attributes beyond HDF5's 64 KB header limit into `name0`, `name1`, ... (saving.py: save_attributes_to_hdf5_group)."""
    if name in attrs and attrs[name] is not None:
        return _strings(attrs[name])
    out, i = [], 0
    while "%s%d" % (name, i) in attrs:
        out += _strings(attrs["%s%d" % (name, i)])
        i += 1
    if not out and "%s0" % name not in attrs:
        raise KeyError("attribute '%s' not found" % name)
    return out


def _read_h5_lite(path):
    """The same walk over a Keras weight file with this package's own HDF5 reader (models/hdf5_lite.py: the "old style" subset
    h5py's default libver writes; no HDF5 library needed)."""
    from . import hdf5_lite
    f = hdf5_lite.File(path)
    g = f["model_weights"] if "model_weights" in f else f.root
    out = {}
    for name in _chunked_attr(g.attrs, "layer_names"):
        layer = g[name]
        weight_names = _chunked_attr(layer.attrs, "weight_names")
        if weight_names:
            out[name] = [layer[w].read() for w in weight_names]
    return out


def _read_h5(path):
    try:
        import h5py
    except ImportError:                                          # the usual case where this package runs: its own reader
        return _read_h5_lite(path)
    out = {}
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        names = [n.decode() if isinstance(n, bytes) else n for n in g.attrs["layer_names"]]
        for name in names:
            wn = [n.decode() if isinstance(n, bytes) else n for n in g[name].attrs["weight_names"]]
            if wn:
                out[name] = [np.asarray(g[name][w]) for w in wn]
    return out


NPZ_CONVERSION = """# run where Keras / h5py exist (the reference's environment); writes the .npz container load_keras_weights() reads
import h5py, numpy as np, sys
with h5py.File(sys.argv[1], "r") as f:
    g = f["model_weights"] if "model_weights" in f else f
    out = {}
    for name in g.attrs["layer_names"]:
        name = name.decode() if isinstance(name, bytes) else name
        for i, w in enumerate(g[name].attrs["weight_names"]):
            out["%s/%d" % (name, i)] = np.asarray(g[name][w.decode() if isinstance(w, bytes) else w])
np.savez(sys.argv[2], **out)
"""


def save_keras_weights_h5(source, path, backend="tensorflow", keras_version="2.2.4"):
    """Write a `{layer_name: [arrays in Keras order]}` dict (or a model: its export_keras_weights) as a Keras weight FILE -- the layout
    `model.save_weights(path)` produces (keras/engine/saving.py save_weights_to_hdf5_group): root attributes `layer_names`, `backend`,
    `keras_version`; one group per layer with the attribute `weight_names` = ["<layer>/kernel:0", "<layer>/bias:0", ...] and the arrays
    as datasets under those paths -- through this package's HDF5 writer (models/hdf5_lite.py; no h5py).  Weight names follow Keras 2:
    Conv2D kernel / bias, BatchNormalization gamma / beta / moving_mean / moving_variance, L2Normalization gamma."""
    from . import hdf5_lite
    weights = source if isinstance(source, dict) else export_keras_weights(source)
    conv, bn = ("kernel:0", "bias:0"), ("gamma:0", "beta:0", "moving_mean:0", "moving_variance:0")
    groups = {}
    for name, arrays in weights.items():
        labels = bn if len(arrays) == 4 else (("gamma:0",) if (len(arrays) == 1 and np.asarray(arrays[0]).ndim == 1) else conv[:len(arrays)])
        names = ["%s/%s" % (name, lab) for lab in labels]
        groups[name] = {"attrs": {"weight_names": np.array([n.encode("utf-8") for n in names])},
                        "groups": {name: {"datasets": {lab: np.asarray(a, dtype=np.float32) for lab, a in zip(labels, arrays)}}}}
    root = {"attrs": {"layer_names": np.array([n.encode("utf-8") for n in weights]), "backend": np.bytes_(backend.encode("utf-8")),
                      "keras_version": np.bytes_(keras_version.encode("utf-8"))},
            "groups": groups}
    hdf5_lite.write(path, root)


def save_keras_weights_npz(source, path):
    """Write a `{layer_name: [arrays in Keras order]}` dict (or a model: its export_keras_weights) as the h5py-free container:
    an .npz whose keys are "<layer name>/<index in Keras' weight_names order>" -- what NPZ_CONVERSION produces from a Keras .h5."""
    weights = source if isinstance(source, dict) else export_keras_weights(source)
    np.savez(path, **{"%s/%d" % (name, i): np.asarray(a) for name, arrays in weights.items() for i, a in enumerate(arrays)})


def _read_npz(path):
    out = {}
    with np.load(path) as z:
        for key in z.files:
            name, _, idx = key.rpartition("/")
            if not name or not idx.isdigit():
                raise ValueError("'{}': keys of a weight .npz read '<layer name>/<index>', got '{}'".format(path, key))
            out.setdefault(name, {})[int(idx)] = z[key]
    return {name: [parts[i] for i in range(len(parts))] for name, parts in out.items()}


def load_keras_weights(model, source, by_name=True, strict_shapes=True):
    """Load Keras-layout weights into `model` (in place).  `source`: a `{layer_name: [arrays]}` dict, the path of an .npz
    container (save_keras_weights_npz / NPZ_CONVERSION) or of a Keras .h5 weight file (h5py if installed, else models/hdf5_lite.py).
    Returns (loaded layer names, model layers absent from the source)."""
    if isinstance(source, str):
        weights = _read_npz(source) if source.endswith(".npz") else _read_h5(source)
    else:
        weights = source
    lm = keras_layer_map(model)
    if not by_name and set(weights) != set(lm):
        raise ValueError("by_name=False needs exactly the model's layers; differing: {}".format(sorted(set(weights) ^ set(lm))))
    loaded, missing = [], []
    with torch.no_grad():
        for name, items in lm.items():
            if name not in weights:
                missing.append(name)
                continue
            arrays = weights[name]
            if len(arrays) != len(items):
                raise ValueError("layer '{}' expects {} weight arrays, the source has {}".format(name, len(items), len(arrays)))
            for (tensor, to_torch, _), a in zip(items, arrays):
                v = np.ascontiguousarray(to_torch(np.asarray(a)))
                if tuple(v.shape) != tuple(tensor.shape):
                    if strict_shapes:
                        raise ValueError("layer '{}': source shape {} does not fit parameter shape {}".format(
                            name, tuple(np.asarray(a).shape), tuple(tensor.shape)))
                    continue
                tensor.copy_(torch.from_numpy(v).to(tensor.dtype))
            loaded.append(name)
    return loaded, missing


def export_keras_weights(model):
    """{Keras layer name: [arrays in Keras layout and order]} (float32 NumPy)."""
    return {name: [np.ascontiguousarray(to_keras(t.detach().float().cpu().numpy())) for t, _, to_keras in items]
            for name, items in keras_layer_map(model).items()}
