#!/usr/bin/env python
"""Converts a Keras HDF5 weight file of the reference (`weights-epoch{N}.h5`, written by `predictive_net.save_weights`,
speechless/net.py:572; e.g. the published checkpoints linked from the reference README) into the `.npz` that
`speechless_amd.net.PredictiveNet.load_weights` reads when `h5py` is not importable next to torch.

Needs only numpy + h5py, so it runs under any interpreter that has them (in this image: /opt/conda/bin/python3.9):

    /opt/conda/bin/python3.9 tools/h5_to_npz.py weights-epoch1234.h5 [out.npz]
    /opt/conda/bin/python3.9 tools/h5_to_npz.py --reverse weights-epoch1234.npz out.h5     # back to the Keras layout

The .npz holds `<layer name>/kernel` (k, Cin, Cout) and `<layer name>/bias` per layer, in the layer order of the file.
"""
import sys

import h5py
import numpy as np


def h5_to_npz(src, dst):
    arrays = {}
    with h5py.File(src, "r") as f:
        root = f["model_weights"] if "model_weights" in f else f
        names = root.attrs.get("layer_names")
        names = [n.decode("utf8") if isinstance(n, bytes) else n for n in names] if names is not None else list(root)
        for name in names:
            group = root[name]
            weight_names = [n.decode("utf8") if isinstance(n, bytes) else n for n in group.attrs.get("weight_names", [])]
            if not weight_names:  # Dropout / Lambda layers carry no weights
                continue
            kernel = [n for n in weight_names if "kernel" in n or n.endswith("W:0") or n.endswith("_W")]
            bias = [n for n in weight_names if "bias" in n or n.endswith("b:0") or n.endswith("_b")]
            w = np.asarray(group[kernel[0]])
            if w.ndim == 4:  # Keras-1 style conv kernels (k, 1, Cin, Cout)
                w = w.reshape(w.shape[0], w.shape[2], w.shape[3])
            arrays[name + "/kernel"] = w
            arrays[name + "/bias"] = np.asarray(group[bias[0]])
    np.savez(dst, **arrays)
    return sorted(arrays)


def npz_to_h5(src, dst):
    data = np.load(src)
    layers = []
    for key in data.files:
        name = key.rsplit("/", 1)[0]
        if name not in layers:
            layers.append(name)
    with h5py.File(dst, "w") as f:
        f.attrs["layer_names"] = [n.encode("utf8") for n in layers]
        for name in layers:
            group = f.create_group(name)
            names = ["{}/kernel:0".format(name), "{}/bias:0".format(name)]
            group.attrs["weight_names"] = [n.encode("utf8") for n in names]
            group.create_dataset(names[0], data=data[name + "/kernel"])
            group.create_dataset(names[1], data=data[name + "/bias"])
    return layers


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--reverse"]
    if not args:
        sys.exit(__doc__)
    if "--reverse" in sys.argv:
        print("\n".join(npz_to_h5(args[0], args[1])))
    else:
        out = args[1] if len(args) > 1 else args[0].rsplit(".", 1)[0] + ".npz"
        print("\n".join(h5_to_npz(args[0], out)))
