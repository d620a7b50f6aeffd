#!/opt/conda/bin/python3.9
"""Real-HDF5-library fixtures for models/hdf5_lite.py (the reference's weights are Keras `.h5` files).

    /opt/conda/bin/python3.9 tests/golden/make_h5_golden.py [out_dir]        (default tests/golden/h5)

This image carries an Anaconda tree with h5py 3.3.0 on the HDF5 C library 1.10.6 (not on the interpreter the package runs on): THIS
script is the only thing that imports it.  It writes small `.h5` files with the real library and a manifest of what the real library
reads back from them (`manifest.json`: every group, dataset shape / dtype, attribute value -- h5py's own walk).  tests/test_host_cpu.py
then reads the committed files with the package's pure-Python reader and compares with the manifest and with `expected_values`
(recomputed from the path, not stored).  Keras itself is absent here (and from /root/reference: a third-party dependency, Keras 2.2.4 in
the reference's environment); the file LAYOUT it produces is restated below from its published saving code
(keras/engine/saving.py: save_weights_to_hdf5_group, save_attributes_to_hdf5_group, _serialize_model), the bytes are the HDF5 library's.

Two attribute styles, because the reference's files are from 2017/18: h5py 2.x stored `bytes` / lists of `bytes` as FIXED-length strings
(numpy 'S' arrays) -- emulated here by handing h5py numpy 'S' values --, h5py 3.x stores them as VARIABLE-length strings (global heap).
"""
import hashlib
import json
import os
import sys
import zlib

import numpy as np

try:                                     # the tests import this module for expected_values / the layer tables on the h5py-less interpreter
    import h5py
except ImportError:
    h5py = None

HDF5_OBJECT_HEADER_LIMIT = 64512        # Keras splits an attribute that would not fit an object header into name0, name1, ...


def expected_values(path, shape, dtype):
    """The values of dataset `path`: a function of the path alone, exact in every dtype used (tests recompute it)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = zlib.crc32(path.encode("utf-8")) % 1000
    v = ((np.arange(n, dtype=np.int64) * 7 + base) % 251 - 125)
    dt = np.dtype(dtype)
    if dt.kind == "f":
        v = v.astype(np.float64) / 8.0
    elif dt.kind == "u":
        v = v + 125
    return v.astype(dt.newbyteorder("=")).reshape(shape)


# ---- the Keras 2.2.4 weight-file layout, restated -----------------------------------------------------------------------------------
def save_attributes(group, name, data, style):
    """A list of byte strings as ONE attribute, or as name0, name1, ... when it would exceed the object-header limit."""
    if style == "h5py2":
        data = np.asarray(data, dtype="S") if len(data) else np.asarray(data)       # fixed-length strings, as h5py 2 stored lists of bytes
        chunks, n = [data], 1
        while any(c.nbytes > HDF5_OBJECT_HEADER_LIMIT for c in chunks):
            n += 1
            chunks = np.array_split(data, n)
        if n > 1:
            for i, c in enumerate(chunks):
                group.attrs["%s%d" % (name, i)] = c
        else:
            group.attrs[name] = data
    else:
        group.attrs[name] = data                                                      # h5py 3: variable-length strings (an empty list: float64[0])


def save_weights(f, layers, style, backend=b"tensorflow", keras_version=b"2.2.4"):
    """layers: [(layer name, [(weight name, shape, dtype), ...])] -- weightless layers get their (empty) group too, as Keras writes them."""
    save_attributes(f, "layer_names", [n.encode("utf8") for n, _ in layers], style)
    f.attrs["backend"] = np.bytes_(backend) if style == "h5py2" else backend
    f.attrs["keras_version"] = np.bytes_(keras_version) if style == "h5py2" else keras_version
    for lname, weights in layers:
        g = f.create_group(lname)
        save_attributes(g, "weight_names", [w.encode("utf8") for w, _, _ in weights], style)
        for wname, shape, dtype in weights:
            d = g.create_dataset(wname, shape, dtype=dtype)                           # 'conv1_1/kernel:0' under group 'conv1_1': a nested group
            val = expected_values(d.name, shape, dtype)
            if len(shape):
                d[:] = val
            else:
                d[()] = val


def conv(name, kh, cin, cout, bias=True):
    w = [("%s/kernel:0" % name, (kh, kh, cin, cout), "float32")]
    return (name, w + ([("%s/bias:0" % name, (cout,), "float32")] if bias else []))


def bn(name, c):
    return (name, [("%s/%s:0" % (name, k), (c,), "float32") for k in ("gamma", "beta", "moving_mean", "moving_variance")])


def ssd300_layers(scale=64, classes=3):
    """Every layer of models/keras_ssd300.py's training-mode graph by its Keras name, in graph order, channel counts divided by `scale`
    (the structure is what is pinned; full-size files are written at test time where this interpreter exists)."""
    c = lambda n: max(1, n // scale)
    L = [("input_1", []), ("identity_layer", []), ("input_mean_normalization", []), ("input_channel_swap", [])]
    trunk = [("conv1_1", 3, 3, 64), ("conv1_2", 3, 64, 64), "pool1", ("conv2_1", 3, 64, 128), ("conv2_2", 3, 128, 128), "pool2",
             ("conv3_1", 3, 128, 256), ("conv3_2", 3, 256, 256), ("conv3_3", 3, 256, 256), "pool3", ("conv4_1", 3, 256, 512),
             ("conv4_2", 3, 512, 512), ("conv4_3", 3, 512, 512), "pool4", ("conv5_1", 3, 512, 512), ("conv5_2", 3, 512, 512),
             ("conv5_3", 3, 512, 512), "pool5", ("fc6", 3, 512, 1024), ("fc7", 1, 1024, 1024), ("conv6_1", 1, 1024, 256), "conv6_padding",
             ("conv6_2", 3, 256, 512), ("conv7_1", 1, 512, 128), "conv7_padding", ("conv7_2", 3, 128, 256), ("conv8_1", 1, 256, 128),
             ("conv8_2", 3, 128, 256), ("conv9_1", 1, 256, 128), ("conv9_2", 3, 128, 256)]
    for t in trunk:
        if isinstance(t, str):
            L.append((t, []))
        else:
            name, k, ci, co = t
            L.append(conv(name, k, 3 if ci == 3 else c(ci), c(co)))
    L.append(("conv4_3_norm", [("conv4_3_norm/conv4_3_norm_gamma:0", (c(512),), "float32")]))
    src = [("conv4_3_norm", 512, 4), ("fc7", 1024, 6), ("conv6_2", 512, 6), ("conv7_2", 256, 6), ("conv8_2", 256, 4), ("conv9_2", 256, 4)]
    for kind, per in (("conf", classes), ("loc", 4)):     # 3 classes in the committed file: the structure is what is pinned
        for s, ci, nb in src:
            L.append(conv("%s_mbox_%s" % (s, kind), 3, c(ci), nb * per))
    for kind in ("priorbox", "conf_reshape", "loc_reshape", "priorbox_reshape"):
        L += [("%s_mbox_%s" % (s, kind), []) for s, _, _ in src]
    L += [("mbox_conf", []), ("mbox_loc", []), ("mbox_priorbox", []), ("mbox_conf_softmax", []), ("predictions", [])]
    return L


def ssd7_layers(scale=8):
    c = lambda n: max(1, n // scale)
    f = [32, 48, 64, 64, 48, 48, 32]
    L = [("input_1", []), ("identity_layer", []), ("input_mean_normalization", []), ("input_stddev_normalization", [])]
    cin = 3
    for i, co in enumerate(f):
        L += [conv("conv%d" % (i + 1), 5 if i == 0 else 3, cin if cin == 3 else c(cin), c(co)), bn("bn%d" % (i + 1), c(co)),
              ("elu%d" % (i + 1), []), ("pool%d" % (i + 1), [])]
        cin = co
    for i in range(4, 8):
        L += [conv("classes%d" % i, 3, c(f[i - 1]), 4 * 6), conv("boxes%d" % i, 3, c(f[i - 1]), 4 * 4)]
    L += [("anchors%d" % i, []) for i in range(4, 8)]
    L += [("classes_concat", []), ("boxes_concat", []), ("anchors_concat", []), ("classes_softmax", []), ("predictions", [])]
    return L


# ---- the files ---------------------------------------------------------------------------------------------------------------------------
def build(out):
    os.makedirs(out, exist_ok=True)
    made = []

    def new(name):
        made.append(name)
        return h5py.File(os.path.join(out, name), "w")                               # libver default ('earliest'): what Keras users get

    with new("ssd300_save_weights_h5py2.h5") as f:                                    # model.save_weights(path), attributes as h5py 2 stored them
        save_weights(f, ssd300_layers(), "h5py2")
    with new("ssd7_save_weights_h5py3.h5") as f:                                      # ... and as h5py 3 stores them (variable-length strings)
        save_weights(f, ssd7_layers(), "h5py3")
    with new("ssd7_model_save_h5py2.h5") as f:                                        # model.save(path): _serialize_model
        f.attrs["keras_version"] = np.bytes_(b"2.2.4")
        f.attrs["backend"] = np.bytes_(b"tensorflow")
        f.attrs["model_config"] = np.bytes_(json.dumps({"class_name": "Model", "config": {"name": "model_1", "layers": ["..."] * 40}}).encode("utf8"))
        save_weights(f.create_group("model_weights"), ssd7_layers(), "h5py2")
        f.attrs["training_config"] = np.bytes_(json.dumps({"optimizer_config": {"class_name": "SGD"}, "loss": "compute_loss"}).encode("utf8"))
        og = f.create_group("optimizer_weights")
        names = [b"SGD/iterations:0", b"training/SGD/Variable:0", b"training/SGD/Variable_1:0"]
        og.attrs["weight_names"] = np.asarray(names, dtype="S")
        for n, shape, dt in ((names[0], (), "int64"), (names[1], (5, 5, 3, 4), "float32"), (names[2], (4,), "float32")):
            d = og.create_dataset(n.decode(), shape, dtype=dt)
            val = expected_values(d.name, shape, dt)
            if len(shape):
                d[:] = val
            else:
                d[()] = val
    with new("structures.h5") as f:                                                   # what a weight file can hold beyond the usual
        g = f.create_group("many_links")                                              # 300 links: a two-level B-tree of group nodes
        for i in range(300):
            d = g.create_dataset("d%03d" % i, (2,), dtype="int16")
            d[:] = expected_values(d.name, (2,), "int16")
        names = [("layer_with_a_long_name_%05d" % i).encode() for i in range(2400)]   # 67 KB of names: layer_names0, layer_names1
        save_attributes(f.create_group("chunked_attribute"), "layer_names", names, "h5py2")
        t = f.create_group("types")
        for name, shape, dt in (("f64", (3, 2), "<f8"), ("f16", (5,), "<f2"), ("i8", (4,), "i1"), ("u16", (4,), "<u2"), ("i32", (2, 2, 2), "<i4"),
                                ("u64", (3,), "<u8"), ("be_f32", (6,), ">f4"), ("be_i16", (3,), ">i2"), ("scalar_f32", (), "<f4"), ("empty", (0, 3), "<f4")):
            d = t.create_dataset(name, shape, dtype=dt)
            val = expected_values(d.name, shape, dt)
            if len(shape) and int(np.prod(shape)):
                d[:] = val
            elif not len(shape):
                d[()] = val
        t.create_dataset("fixed_strings", data=np.asarray([b"alpha", b"be", b"gamma_delta"], dtype="S"))
        t.create_dataset("vlen_strings", data=[b"alpha", b"", b"gamma_delta"], dtype=h5py.string_dtype("ascii"))
        t.create_dataset("never_written", (4,), dtype="float32")                      # late allocation: no storage, reads as zeros in h5py
        space = h5py.h5s.create_simple((6,))
        plist = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        plist.set_layout(h5py.h5d.COMPACT)                                            # data inside the object header
        did = h5py.h5d.create(t.id, b"compact_f32", h5py.h5t.IEEE_F32LE, space, plist)
        did.write(h5py.h5s.ALL, h5py.h5s.ALL, expected_values("/types/compact_f32", (6,), "<f4"))
        t.create_dataset("chunked", data=expected_values("/types/chunked", (8, 8), "<f4"), chunks=(4, 4))     # the reader must refuse it clearly
        a = f.create_group("attributes")
        a.attrs["int_scalar"] = np.int32(-7)
        a.attrs["float_vector"] = np.asarray([1.5, -2.25, 1e-3], np.float64)
        a.attrs["fixed_bytes"] = np.bytes_(b"tensorflow")
        a.attrs["vlen_bytes"] = b"tensorflow"
        a.attrs["vlen_str"] = "2.2.4"
        a.attrs["vlen_list"] = [b"a/kernel:0", b"a/bias:0"]
        a.attrs["empty_list"] = []
        a.attrs["matrix"] = np.arange(6, dtype=np.int64).reshape(2, 3)
        for i in range(30):                                                           # enough attributes to need continuation blocks
            a.attrs["filler_%02d" % i] = np.bytes_(("value %d " % i).encode() * 3)

    # ---- the manifest: what the REAL library reads back --------------------------------------------------------------------------------
    def jsonable(v):
        if isinstance(v, bytes):
            return {"bytes": v.decode("latin-1")}
        if isinstance(v, str):
            return {"str": v}
        if isinstance(v, np.ndarray):
            if v.dtype.kind in "SO":
                strings = [x.decode("latin-1") if isinstance(x, bytes) else str(x) for x in v.reshape(-1).tolist()]
                if len(strings) > 200:                                                # long lists by digest
                    return {"strings_sha1": hashlib.sha1("\n".join(strings).encode("latin-1")).hexdigest(), "count": len(strings), "shape": list(v.shape)}
                return {"strings": strings, "shape": list(v.shape)}
            return {"array": v.reshape(-1).tolist(), "shape": list(v.shape), "dtype": v.dtype.newbyteorder("=").str.lstrip("=<>|")}
        if isinstance(v, np.generic):
            return jsonable(np.asarray(v))
        return {"value": v}

    manifest = {"_generator": "tests/golden/make_h5_golden.py", "_h5py": h5py.__version__, "_hdf5": h5py.version.hdf5_version, "files": {}}
    for name in made:
        entry = {"groups": {}, "datasets": {}}
        with h5py.File(os.path.join(out, name), "r") as f:
            def visit(path, obj):
                attrs = {k: jsonable(obj.attrs[k]) for k in obj.attrs}
                if isinstance(obj, h5py.Dataset):
                    kind = "chunked" if obj.chunks else ("vlen" if h5py.check_string_dtype(obj.dtype) and obj.dtype.kind == "O" else "plain")
                    info = {"shape": list(obj.shape), "dtype": obj.dtype.str if obj.dtype.kind != "O" else "O", "kind": kind, "attrs": attrs}
                    if obj.dtype.kind in "SO":
                        info["strings"] = [x.decode("latin-1") if isinstance(x, bytes) else str(x) for x in np.asarray(obj[()]).reshape(-1).tolist()]
                    elif path.endswith("never_written"):
                        info["zeros"] = True
                    entry["datasets"]["/" + path] = info
                else:
                    entry["groups"]["/" + path] = {"keys": sorted(obj.keys()), "attrs": attrs}
            entry["groups"]["/"] = {"keys": sorted(f.keys()), "attrs": {k: jsonable(f.attrs[k]) for k in f.attrs}}
            f.visititems(visit)
            # every numeric dataset holds expected_values(its path): checked here with the real library, recomputed by the tests
            for path, info in entry["datasets"].items():
                if info["kind"] == "plain" and info["dtype"] not in ("O",) and not info["dtype"].startswith("|S") and "zeros" not in info:
                    got = f[path][()]
                    assert np.array_equal(np.asarray(got).astype(np.float64), expected_values(path, tuple(info["shape"]), info["dtype"]).astype(np.float64)), path
        entry["bytes"] = os.path.getsize(os.path.join(out, name))
        manifest["files"][name] = entry
    with open(os.path.join(out, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=0, sort_keys=True)
    return made, manifest


def write_full_size(path, which):
    """A full-size weight file of the named model in Keras' layout, by the real library (tests: file -> this package's model)."""
    with h5py.File(path, "w") as f:
        save_weights(f, ssd300_layers(1, 21) if which == "ssd300" else ssd7_layers(1), "h5py2")


def dump_with_h5py(path):
    """What the real library reads from `path` (a file this package's writer produced): JSON on stdout."""
    out = {"datasets": {}, "attrs": {}}
    with h5py.File(path, "r") as f:
        def visit(name, obj):
            out["attrs"]["/" + name] = {k: (np.asarray(obj.attrs[k]).astype("S").tolist() if np.asarray(obj.attrs[k]).dtype.kind in "SO" else np.asarray(obj.attrs[k]).tolist()) for k in obj.attrs}
            if isinstance(obj, h5py.Dataset):
                a = obj[()]
                out["datasets"]["/" + name] = {"shape": list(obj.shape), "dtype": obj.dtype.str, "sha1": hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()}
        out["attrs"]["/"] = {k: (np.asarray(f.attrs[k]).astype("S").tolist() if np.asarray(f.attrs[k]).dtype.kind in "SO" else np.asarray(f.attrs[k]).tolist()) for k in f.attrs}
        f.visititems(visit)
    def fix(o):
        if isinstance(o, bytes):
            return o.decode("latin-1")
        if isinstance(o, list):
            return [fix(x) for x in o]
        if isinstance(o, dict):
            return {k: fix(v) for k, v in o.items()}
        return o
    print(json.dumps(fix(out)))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--full":
    write_full_size(sys.argv[3], sys.argv[2])
elif __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--dump":
    dump_with_h5py(sys.argv[2])
elif __name__ == "__main__":
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5")
    files, man = build(out_dir)
    for n in files:
        print("%-40s %8d bytes  %4d groups %4d datasets" % (n, man["files"][n]["bytes"], len(man["files"][n]["groups"]), len(man["files"][n]["datasets"])))
