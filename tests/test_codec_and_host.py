"""CPU tests of the host logic at the boundary: label codec (pinned by the REFERENCE's own module/tests through
tests/golden/codec_golden.json), result objects, batch packing, C-ABI surface."""
import json
import re
from pathlib import Path

import numpy as np
import pytest

from speechless_amd.grapheme_encoding import (CtcGraphemeEncoding, english_frequent_characters,
                                              german_frequent_characters)

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = json.loads((ROOT / "tests" / "golden" / "codec_golden.json").read_text(encoding="utf8"))


def test_alphabets_match_reference():
    assert english_frequent_characters == GOLDEN["english_frequent_characters"]
    assert german_frequent_characters == GOLDEN["german_frequent_characters"]


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: c["alphabet"])
def test_codec_matches_reference_outputs(case):
    chars = GOLDEN[case["alphabet"] + "_frequent_characters"]
    g = CtcGraphemeEncoding(chars)
    assert g.grapheme_set_size == case["grapheme_set_size"]
    assert g.ctc_blank == case["ctc_blank"]
    for label, encoded in case["encode"].items():
        assert g.encode(label) == encoded
        assert g.decode_graphemes(g.encode(label), merge_repeated=False) == label  # test_grapheme_encoding.py:10-13
    batch = g.encode_label_batch(case["encode_label_batch"]["labels"])
    assert batch.dtype == np.int32 and batch.tolist() == case["encode_label_batch"]["result"]
    for d in case["decode_graphemes"]:
        assert g.decode_graphemes(d["graphemes"], merge_repeated=d["merge_repeated"]) == d["result"]
    d = case["decode_prediction_batch"]
    assert g.decode_prediction_batch(np.array(d["predictions"]), d["prediction_lengths"]) == d["result"]
    assert case["test_encode_batch"]["result"] == ["abc", "ab"]  # test_grapheme_encoding.py:20-31


def test_reference_collapse_example():
    g = CtcGraphemeEncoding(english_frequent_characters)
    graphemes = g.encode("sssshhhheeeee      wasn't thre") + [g.ctc_blank] + g.encode("eeeeee")
    assert g.decode_graphemes(graphemes) == "she wasn't three"  # test_grapheme_encoding.py:15-18


def test_unknown_character_raises_value_error():
    with pytest.raises(ValueError):
        CtcGraphemeEncoding(english_frequent_characters).encode("ä")


def test_edit_distance_and_result_objects():
    from speechless_amd.net import ExpectationVsPrediction, ExpectationsVsPredictions, edit_distance
    assert edit_distance("kitten", "sitting") == 3
    assert edit_distance([], ["a"]) == 1
    assert edit_distance("abc", "abc") == 0
    r = ExpectationVsPrediction(expected="she wasn't three", predicted="she wasnt tree", loss=1.5)
    assert r.letter_error_count == 2 and r.word_error_count == 2
    assert abs(r.letter_error_rate - 2 / 16) < 1e-12 and abs(r.word_error_rate - 2 / 3) < 1e-12
    assert "loss: 1.50" in str(r)
    rs = ExpectationsVsPredictions([r, ExpectationVsPrediction("a b", "a b", 0.5)])
    assert rs.average_loss == 1.0 and "Average over 2 examples" in rs.summary_line()


def test_same_padding_and_plan_geometry():
    from speechless_amd.engine import LayerPlan, same_padding, wav2letter_layer_specs
    assert same_padding(1000, 48, 2) == (500, 23, 23)
    assert same_padding(999, 48, 2) == (500, 23, 24)
    specs = wav2letter_layer_specs(128, 29)
    p0 = LayerPlan(0, specs[0], 128, 256, 0, 0)
    assert (p0.taps_view, p0.cin_view, p0.pad_left) == (24, 256, 23)
    p9 = LayerPlan(8, specs[8], 256, 2048, 0, 0)
    assert (p9.taps_view, p9.pad_left, p9.pad_right) == (32, 15, 16)


def test_header_symbols_are_exported_and_bound():
    """The C-ABI library loads and exports every symbol include/speechless_hip.h declares (no compute calls)."""
    from speechless_amd import _lib
    from speechless_amd.build import build
    build()
    header = (ROOT / "include" / "speechless_hip.h").read_text()
    declared = set(re.findall(r"\b(sl_[a-z0-9_]+)\s*\(", header))
    declared -= {"sl_status", "sl_dtype", "sl_epilogue", "sl_conv_geom"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    library = _lib.HipLibrary()
    assert library.raw("sl_version")() == 1
    # argument validation works without a GPU: a bad geometry is rejected before any launch
    geom = _lib.ConvGeom()
    rc = library.raw("sl_conv1d_nt")(1, 1, None, None, 1, geom, 0, 0, 0, 0, None, 0, None)
    assert rc == -1 and "must be positive" in library.last_error()


def test_host_header_symbols_are_exported_and_bound():
    """libspeechless_host.so exports every symbol include/speechless_host.h declares and the ctypes table binds exactly
    those (the sources include the header, so the compiler has checked the signatures against the definitions)."""
    from speechless_amd import _host_lib
    from speechless_amd.build import build_host
    build_host()
    header = (ROOT / "include" / "speechless_host.h").read_text()
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sl_host_[a-z0-9_]+)\s*\(", code))
    assert declared == set(_host_lib.HOST_SIGNATURES), declared ^ set(_host_lib.HOST_SIGNATURES)
    lib = _host_lib.host_lib()
    assert lib.sl_host_version() == 1
    assert lib.sl_host_pack_batch(None, None, 0, 0, 0, 0, None, 1) == -1  # argument validation, no work
    for src in ("pack_batch.cpp", "beam_search.cpp"):
        assert "speechless_host.h" in (ROOT / "speechless_amd" / "csrc_host" / src).read_text()


def test_product_code_never_imports_the_oracle():
    for path in (ROOT / "speechless_amd").rglob("*.py"):
        text = path.read_text()
        assert "import oracle" not in text and "from oracle" not in text, path


def test_batchnorm_folds_into_the_fused_conv_bias_relu_epilogue():
    """relu(BN(conv(x, W) + b)) == relu(conv(x, W') + b') with the folded parameters (speechless_amd/fold.py)."""
    from oracle import w2l_oracle as o
    from speechless_amd.fold import fold_batchnorm_into_conv
    rng = np.random.RandomState(3)
    x = rng.randn(2, 40, 6)
    w = rng.randn(5, 6, 9) * 0.2
    b = rng.randn(9) * 0.1
    gamma, beta = rng.uniform(0.5, 1.5, 9), rng.randn(9) * 0.1
    mean, var, eps = rng.randn(9) * 0.3, rng.uniform(0.2, 2.0, 9), 1e-3
    z = o.conv1d_preactivation(x, w, b, 1)
    want = np.maximum(gamma * (z - mean) / np.sqrt(var + eps) + beta, 0)
    w2, b2 = fold_batchnorm_into_conv(w, b, gamma, beta, mean, var, eps)
    got = np.maximum(o.conv1d_preactivation(x, w2, b2, 1), 0)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    with pytest.raises(ValueError):
        fold_batchnorm_into_conv(w, b, gamma[:3], beta, mean, var)


def test_keras_h5_checkpoints_convert_to_the_npz_the_net_loads(tmp_path):
    """tools/h5_to_npz.py (numpy only, through speechless_amd/h5lite.py) turns a Keras-layout weight file into the flat
    .npz form and back; the committed h5py-written fixture converts to its expected arrays."""
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    tool = str(root / "tools" / "h5_to_npz.py")
    rng = np.random.RandomState(0)
    layers = {"striding_conv": (48, 128, 250), "inner_conv_1": (7, 250, 250), "output_conv": (1, 2000, 29)}
    npz = tmp_path / "weights-epoch3.npz"
    np.savez(npz, **{n + "/kernel": rng.randn(*sh).astype(np.float32) for n, sh in layers.items()},
             **{n + "/bias": rng.randn(sh[2]).astype(np.float32) for n, sh in layers.items()})
    h5 = tmp_path / "weights-epoch3.h5"
    assert subprocess.run([sys.executable, tool, "--reverse", str(npz), str(h5)], capture_output=True).returncode == 0
    back = tmp_path / "back.npz"
    res = subprocess.run([sys.executable, tool, str(h5), str(back)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    a, b = np.load(npz), np.load(back)
    assert sorted(a.files) == sorted(b.files)
    for key in a.files:
        assert np.array_equal(a[key], b[key])
    res = subprocess.run([sys.executable, tool, str(root / "tests" / "golden" / "keras_model_toy.h5"),
                          str(tmp_path / "toy.npz")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    got, want = np.load(tmp_path / "toy.npz"), np.load(root / "tests" / "golden" / "keras_h5_expected.npz")
    assert sorted(got.files) == sorted(want.files) and all(np.array_equal(got[k], want[k]) for k in want.files)


def test_native_batch_packer_matches_the_numpy_packing():
    """libspeechless_host.so:sl_host_pack_batch == the reference's zero-padding loop (net.py:583-586), f64 and f32."""
    from speechless_amd.pipeline import pack_spectrograms
    rng = np.random.RandomState(5)
    for dtype in (np.float64, np.float32):
        specs = [rng.randn(int(t), 37).astype(dtype) for t in (50, 1, 33, 50, 17)]
        want = np.zeros((5, 50, 37), dtype=np.float32)
        for row, s in zip(want, specs):
            row[:s.shape[0]] = s
        got = np.full((5, 50, 37), 9.0, dtype=np.float32)
        pack_spectrograms(specs, got, n_threads=3)
        assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        pack_spectrograms([rng.randn(60, 37)], np.zeros((1, 50, 37), dtype=np.float32))


def test_bench_line_is_compact_enough_for_the_driver():
    """the round-4 line (20 KB) could not be extracted by the driver.  The last stdout line is now a
    summary of the full detail; fed with the round-4 detail it must stay far below the driver's limit and keep the
    contract's keys, `roofline` and `cpu_baseline`."""
    import json
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    import bench
    detail = json.loads((root / "profiles" / "r04_bench.json").read_text().strip().splitlines()[-1])
    detail["_config_id"] = 3
    detail["data_parallel"] = {"world_size": 8, "backend": "nccl", "rccl_version": "2.26.6", "bucket_bytes": [1, 2, 3],
                               "sharded_optimizer": False, "reduced_gradients_and_weights_identical_on_all_ranks": True,
                               "allreduce_alone_ms": 1.0, "allreduce_busbw_GBps": 100.0, "step_ms_with_allreduce": 2.2,
                               "step_ms_without_allreduce": 2.1, "exposed_communication_ms": 0.1, "note": "x" * 500}
    line = bench.compact_line(detail, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < 4000, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind"):
        assert key in line["cpu_baseline"], key
    assert "workload" in line["config"]
    assert len(json.dumps(line["data_parallel"])) < 2000
    # the tracked rocprofv3 average printed beside the live duration comes from the config-3 bf16 statistics, not from the files
    # of other configurations / arithmetic paths kept next to them
    ms, calls, source = bench.tracked_rocprof_average("wgrad_tn_ilv_kernel")
    import re
    assert re.fullmatch(r"profiles/r\d+[a-z]?_kernel_stats\.csv", source) and 0.15 < ms < 0.30 and calls > 0, (ms, calls, source)
