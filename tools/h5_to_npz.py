#!/usr/bin/env python
"""Converts a Keras HDF5 weight file of the reference (`weights-epoch{N}.h5`, written by `predictive_net.save_weights`,
speechless/net.py:572; e.g. the published checkpoints linked from the reference README) into the flat `.npz` form
(`<layer name>/kernel` (k, Cin, Cout), `<layer name>/bias`) and back.  Since round 3 `PredictiveNet.load_weights` reads the
.h5 itself (speechless_amd/h5lite.py); this tool remains for inspecting checkpoints with plain numpy.  numpy only.

    python tools/h5_to_npz.py weights-epoch1234.h5 [out.npz]
    python tools/h5_to_npz.py --reverse weights-epoch1234.npz out.h5     # back to the Keras layout
"""
import importlib.util
import sys
from pathlib import Path

import numpy as np

_spec = importlib.util.spec_from_file_location(
    "h5lite", str(Path(__file__).resolve().parent.parent / "speechless_amd" / "h5lite.py"))  # (no package import: no torch)
h5lite = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(h5lite)


def h5_to_npz(src, dst):
    arrays = {}
    for name, weights in h5lite.read_keras_weights(src):
        kernel = [v for n, v in weights.items() if "kernel" in n or n.endswith("W:0") or n.endswith("_W")]
        bias = [v for n, v in weights.items() if "bias" in n or n.endswith("b:0") or n.endswith("_b")]
        w = np.asarray(kernel[0])
        if w.ndim == 4:  # Keras-1 style conv kernels (k, 1, Cin, Cout)
            w = w.reshape(w.shape[0], w.shape[2], w.shape[3])
        arrays[name + "/kernel"] = w
        arrays[name + "/bias"] = np.asarray(bias[0])
    np.savez(dst, **arrays)
    return sorted(arrays)


def npz_to_h5(src, dst):
    data = np.load(src)
    layers = []
    for key in data.files:
        name = key.rsplit("/", 1)[0]
        if name not in layers:
            layers.append(name)
    h5lite.write_keras_weights(dst, [(name, [("{}/kernel:0".format(name), data[name + "/kernel"]),
                                             ("{}/bias:0".format(name), data[name + "/bias"])]) for name in layers])
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
