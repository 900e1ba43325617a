#!/usr/bin/env python
"""Writes the Keras-HDF5 fixtures of tests/golden/ with the REAL HDF5 library (h5py), in the two layouts Keras 2.0 produces
for the reference's checkpoints (speechless/net.py:209-212 load_weights, :558-572 save_weights):

    keras_weights_toy.h5   what `model.save_weights()` writes: root attributes `layer_names` / `backend` / `keras_version`,
                           one group per layer (also for the weight-less Dropout / Lambda layers) with a `weight_names`
                           attribute, the arrays below it as `<layer>/<layer>/kernel:0`, `<layer>/<layer>/bias:0`
    keras_model_toy.h5     what `model.save()` writes: the same tree under a `model_weights` group, beside a `model_config`
                           attribute and an `optimizer_weights` group
    keras_h5_expected.npz  the arrays themselves (`<layer>/kernel`, `<layer>/bias`)

Run under an interpreter that has numpy + h5py (in the build image: /opt/conda/bin/python3.9).  The fixtures are DATA --
pinned inputs for speechless_amd/h5lite.py, the dependency-free reader / writer the product uses where h5py is absent.
"""
import json
import sys
from pathlib import Path

import h5py
import numpy as np

HERE = Path(__file__).resolve().parent
# the reference topology (net.py:291-341) with toy channel counts: (name, kernel size, cin, cout)
LAYERS = [("striding_conv", 6, 4, 6)] + [("inner_conv_{}".format(i), 3, 6, 6) for i in (1, 2)] + \
    [("big_conv_1", 4, 6, 8), ("big_conv_2", 1, 8, 8), ("output_conv", 1, 8, 5)]
WEIGHTLESS = ["dropout_1", "loss_lambda"]  # Keras writes a group with an empty weight_names list for these


def arrays():
    rng = np.random.RandomState(7)
    return {name: (rng.randn(k, cin, cout).astype(np.float32), rng.randn(cout).astype(np.float32))
            for name, k, cin, cout in LAYERS}


def write_weight_tree(root, data):
    names = [LAYERS[0][0], WEIGHTLESS[0]] + [l[0] for l in LAYERS[1:]] + [WEIGHTLESS[1]]
    root.attrs["layer_names"] = [n.encode("utf8") for n in names]
    root.attrs["backend"] = "tensorflow".encode("utf8")
    root.attrs["keras_version"] = "2.0.2".encode("utf8")
    for name in names:
        group = root.create_group(name)
        if name in data:
            weight_names = ["{}/kernel:0".format(name), "{}/bias:0".format(name)]
            group.attrs["weight_names"] = [n.encode("utf8") for n in weight_names]
            for weight_name, value in zip(weight_names, data[name]):
                dataset = group.create_dataset(weight_name, value.shape, dtype=value.dtype)  # -> <layer>/<layer>/kernel:0
                dataset[...] = value
        else:
            group.attrs["weight_names"] = np.zeros((0,), dtype="S1")


def main():
    data = arrays()
    with h5py.File(str(HERE / "keras_weights_toy.h5"), "w") as f:
        write_weight_tree(f, data)
    with h5py.File(str(HERE / "keras_model_toy.h5"), "w") as f:
        f.attrs["keras_version"] = "2.0.2".encode("utf8")
        f.attrs["backend"] = "tensorflow".encode("utf8")
        f.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {"name": "toy"}}).encode("utf8")
        write_weight_tree(f.create_group("model_weights"), data)
        opt = f.create_group("optimizer_weights")
        opt.attrs["weight_names"] = [b"Adam/iterations:0"]
        opt.create_dataset("Adam/iterations:0", data=np.array(12, dtype=np.int64))
    np.savez(str(HERE / "keras_h5_expected.npz"), **{n + "/kernel": w for n, (w, _) in data.items()},
             **{n + "/bias": b for n, (_, b) in data.items()})
    print("wrote", [p.name for p in sorted(HERE.glob("keras_*"))], "with h5py", h5py.__version__, "HDF5",
          h5py.version.hdf5_version)


def check(path):
    """`make_keras_h5_fixture.py --check file.h5`: lists what the real HDF5 library sees in a file (used by the tests to
    validate what speechless_amd/h5lite.py WROTE); prints JSON."""
    out = {"attrs": {}, "datasets": {}}
    with h5py.File(path, "r") as f:
        def to_list(v):
            v = np.asarray(v)
            return [x.decode("utf8") if isinstance(x, bytes) else x for x in v.reshape(-1).tolist()]
        out["attrs"]["/"] = {k: to_list(v) for k, v in f.attrs.items()}

        def visit(name, obj):
            out["attrs"]["/" + name] = {k: to_list(v) for k, v in obj.attrs.items()}
            if isinstance(obj, h5py.Dataset):
                value = np.asarray(obj)
                out["datasets"]["/" + name] = {"shape": list(value.shape), "dtype": str(value.dtype),
                                               "sum": float(value.astype(np.float64).sum()),
                                               "first": value.reshape(-1)[:4].astype(np.float64).tolist()}
        f.visititems(visit)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--check":
        check(sys.argv[2])
    else:
        main()
