"""speechless_amd/h5lite.py -- the dependency-free HDF5 reader / writer behind PredictiveNet.save_weights / load_weights
(reference: Keras HDF5 checkpoints, speechless/net.py:209-212, 558-572) -- against the REAL HDF5 library:
  * reading: the committed fixtures tests/golden/keras_{weights,model}_toy.h5 were written by h5py 3.3 / HDF5 1.10.6
    (tests/golden/make_keras_h5_fixture.py) in the layouts of Keras 2.0's save_weights() and model.save();
  * writing: what h5lite writes is read back by h5py under an interpreter that has it (skipped where none exists)."""
import importlib.util
import json
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"


def _h5lite():
    # loaded from its file: the module must not need the package (torch) around it
    spec = importlib.util.spec_from_file_location("h5lite_standalone", str(ROOT / "speechless_amd" / "h5lite.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def _h5py_interpreter():
    py = shutil.which("python3.9", path="/opt/conda/bin") or shutil.which("python3.9")
    if py is None or subprocess.run([py, "-c", "import h5py"], capture_output=True).returncode != 0:
        return None
    return py


@pytest.mark.parametrize("name", ["keras_weights_toy.h5", "keras_model_toy.h5"])
def test_reads_files_written_by_the_hdf5_library(name):
    h5 = _h5lite()
    expected = np.load(str(GOLDEN / "keras_h5_expected.npz"))
    root = h5.read(GOLDEN / name)
    assert root.attrs["keras_version"] == "2.0.2" and root.attrs["backend"] == "tensorflow"
    tree = root["model_weights"] if "model_weights" in root else root
    assert tree.attrs["layer_names"] == ["striding_conv", "dropout_1", "inner_conv_1", "inner_conv_2", "big_conv_1",
                                         "big_conv_2", "output_conv", "loss_lambda"]
    assert tree["dropout_1"].attrs["weight_names"] == [] and len(tree["dropout_1"]) == 0
    layers = h5.read_keras_weights(GOLDEN / name)
    assert [n for n, _ in layers] == ["striding_conv", "inner_conv_1", "inner_conv_2", "big_conv_1", "big_conv_2",
                                      "output_conv"]  # file order, weight-less layers left out
    for layer, weights in layers:
        assert list(weights) == ["{}/kernel:0".format(layer), "{}/bias:0".format(layer)]
        assert np.array_equal(weights[layer + "/kernel:0"], expected[layer + "/kernel"])
        assert np.array_equal(weights[layer + "/bias:0"], expected[layer + "/bias"])
        assert weights[layer + "/kernel:0"].dtype == np.float32
    if "optimizer_weights" in root:
        assert int(root["optimizer_weights"]["Adam"]["iterations:0"].value) == 12


def test_round_trip_and_unsupported_features_are_named(tmp_path):
    h5 = _h5lite()
    rng = np.random.RandomState(1)
    layers = [("a_conv", [("a_conv/kernel:0", rng.randn(3, 4, 5).astype(np.float32)),
                          ("a_conv/bias:0", rng.randn(5).astype(np.float32))]),
              ("z", [("z/kernel:0", rng.randn(1, 5, 2)), ("z/bias:0", np.arange(2, dtype=np.int64))])]
    path = tmp_path / "w.h5"
    h5.write_keras_weights(path, layers)
    back = h5.read_keras_weights(path)
    assert [n for n, _ in back] == ["a_conv", "z"]
    for (_, want), (_, got) in zip(layers, back):
        for name, value in want:
            assert np.array_equal(got[name], value) and got[name].dtype == np.asarray(value).dtype
    with pytest.raises(ValueError, match="not an HDF5 file"):
        bad = tmp_path / "bad.h5"
        bad.write_bytes(b"PK\x03\x04" + b"\x00" * 200)
        h5.read(bad)
    data = bytearray(path.read_bytes())
    data[8] = 2  # superblock version 2 (libver='latest'): refused by name, not misread
    (tmp_path / "v2.h5").write_bytes(bytes(data))
    with pytest.raises(h5.H5Unsupported, match="superblock version 2"):
        h5.read(tmp_path / "v2.h5")


def test_hdf5_library_reads_what_h5lite_writes(tmp_path):
    py = _h5py_interpreter()
    if py is None:
        pytest.skip("no interpreter with h5py in this image")
    h5 = _h5lite()
    rng = np.random.RandomState(2)
    # the real topology's names and (small) shapes, 40 layers' worth of links in one group to exercise the node sizes
    layers = [("layer_{:02d}".format(i), [("layer_{:02d}/kernel:0".format(i), rng.randn(2, 3, 4).astype(np.float32)),
                                           ("layer_{:02d}/bias:0".format(i), rng.randn(4).astype(np.float32))])
              for i in range(40)]
    path = tmp_path / "w.h5"
    h5.write_keras_weights(path, layers)
    res = subprocess.run([py, str(GOLDEN / "make_keras_h5_fixture.py"), "--check", str(path)], capture_output=True,
                         text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    seen = json.loads(res.stdout)
    assert seen["attrs"]["/"]["layer_names"] == [n for n, _ in layers]
    assert seen["attrs"]["/"]["keras_version"] == ["2.0.2"]
    for name, weights in layers:
        assert seen["attrs"]["/" + name]["weight_names"] == [w for w, _ in weights]
        for weight_name, value in weights:
            got = seen["datasets"]["/{}/{}".format(name, weight_name)]
            assert got["shape"] == list(value.shape) and got["dtype"] == "float32"
            assert abs(got["sum"] - float(value.astype(np.float64).sum())) < 1e-9
            assert got["first"] == value.reshape(-1)[:4].astype(np.float64).tolist()


def test_fixtures_are_what_the_generating_script_writes(tmp_path):
    """the committed fixtures are reproducible from tests/golden/make_keras_h5_fixture.py (arrays compared, not bytes:
    HDF5 files carry modification times)"""
    py = _h5py_interpreter()
    if py is None:
        pytest.skip("no interpreter with h5py in this image")
    work = tmp_path / "golden"
    work.mkdir()
    shutil.copy(str(GOLDEN / "make_keras_h5_fixture.py"), str(work))
    assert subprocess.run([py, str(work / "make_keras_h5_fixture.py")], capture_output=True).returncode == 0
    h5 = _h5lite()
    for name in ("keras_weights_toy.h5", "keras_model_toy.h5"):
        a, b = h5.read_keras_weights(work / name), h5.read_keras_weights(GOLDEN / name)
        assert [n for n, _ in a] == [n for n, _ in b]
        for (_, wa), (_, wb) in zip(a, b):
            assert all(np.array_equal(wa[k], wb[k]) for k in wa)
