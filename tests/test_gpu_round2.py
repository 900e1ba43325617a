"""GPU parity tests added in round 2 (run with -m gpu on an MI355X): committed golden fixtures through the HIP path,
configuration 5 at its full length, bf16 gradients at 1000 frames against the CPU path, a long bf16-vs-fp32 training
trajectory, transfer-learning weight surgery, BatchNorm folding through the fused epilogue, optimizer-state resume,
dropout behind ELU, variable-length batches in shared buffers.  Helpers come from test_gpu_parity.py."""
import ctypes
from pathlib import Path

import numpy as np
import pytest

from oracle import w2l_oracle as o
from test_gpu_parity import (_report, layer_activation, make_case, make_engine, rel_l2, run_loss_and_grads,
                             synthetic_examples, weights64)

pytestmark = pytest.mark.gpu

GOLDEN = Path(__file__).resolve().parent / "golden"
TOY_SIZES = dict(main_filter_count=6, out_filter_count=8, striding_kernel=6, inner_kernel=3, big_kernel=4,
                 inner_count=2)


# ------------------------------------------------------------------------------------------ committed fixtures
@pytest.mark.parametrize("name", ["toy", "real"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_committed_stack_fixture_through_the_hip_path(name, dtype):
    """tests/golden/stack_golden.npz (float64 oracle outputs committed in round 1: probabilities, per-utterance losses,
    decoded indices, gradient norms / slices / bias gradients of a toy stack and of the real topology) against the HIP
    kernels -- the fixture, not a freshly computed oracle value, is the expectation."""
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    g = np.load(str(GOLDEN / "stack_golden.npz"))
    f, k, sizes = (4, 5, TOY_SIZES) if name == "toy" else (128, 29, {})
    ospecs = o.layer_specs(f, k, **sizes)
    weights = o.glorot_uniform_weights(ospecs, seed=2, dtype=np.float64)
    weights = [(w.astype(np.float32), g["{}/bias{}".format(name, i)].astype(np.float32))
               for i, (w, _) in enumerate(weights)]
    eng = Engine(wav2letter_layer_specs(f, k, **sizes), k, dtype=dtype)
    eng.set_weights(weights)
    case = dict(x=g[name + "/x"], labels=g[name + "/labels"], label_lengths=g[name + "/label_lengths"],
                prediction_lengths=g[name + "/prediction_lengths"])
    losses, grads = run_loss_and_grads(eng, case)
    probs = eng.cur.probs.cpu().numpy()
    if dtype == "f32":
        assert np.abs(probs - g[name + "/probs"]).max() < 2e-6
        np.testing.assert_allclose(losses, g[name + "/losses"], rtol=1e-5)
        decoded, _ = eng.greedy_decode(case["prediction_lengths"])
        for i, d in enumerate(decoded):
            assert d == list(g[name + "/decoded"][i][:g[name + "/decoded_lengths"][i]])
        for i, (dw, db) in enumerate(grads):
            np.testing.assert_allclose(np.linalg.norm(dw.astype(np.float64)), g["{}/dw_norm{}".format(name, i)],
                                       rtol=2e-4)
            assert rel_l2(db, g["{}/db{}".format(name, i)]) < 2e-4, i
            want = g["{}/dw_slice{}".format(name, i)]
            assert np.abs(dw[0, :4, :4] - want).max() < 2e-4 * np.abs(dw[0]).max() + 1e-9, i
    else:
        np.testing.assert_allclose(losses, g[name + "/losses"], rtol=2e-3)
        assert np.abs(probs - g[name + "/probs"]).max() < 3e-3


# ------------------------------------------------------------------------------------------ configuration 5, full length
def _long_form_case(seed=3, batch=8, bins=257):
    rng = np.random.RandomState(seed)
    lengths = np.sort(rng.randint(2000, 8001, size=batch))[::-1].copy()
    lengths[0] = 8000  # the longest utterance of the configuration: T' = 4000, sixteen 256-row time tiles
    x = np.zeros((batch, int(lengths.max()), bins), dtype=np.float32)
    for i, n in enumerate(lengths):
        x[i, :n] = np.random.RandomState(500 + i).randn(int(n), bins)
    lab_len = [int(min(200, rng.randint(20, n // 4 + 1))) for n in lengths]
    lab_len[0] = 200
    labels = o.pack_label_batch([list(rng.randint(0, 28, size=n)) for n in lab_len])
    pred_len = [int(n) // 2 for n in lengths]
    return x, lengths, labels, lab_len, pred_len


def test_long_form_power_spectrogram_batch_at_full_length():
    """BASELINE config 5 at its defining size: 8 utterances of 257-bin power spectrograms, 2000..8000 frames (the
    longest one 8000: a 4000-frame lattice, 16 time tiles), labels up to 200, zero-padded to the batch maximum like the
    reference packs them (net.py:578-587).  Against the torch-CPU fp32 realisation of the oracle: per-utterance loss
    within 1e-3 (bf16) / 2e-5 (fp32), greedy decode of the fp32 path bit-exact; frames that are not scored
    (t >= prediction_length) receive exactly zero gradient."""
    import torch
    from oracle import w2l_torch_cpu as tc
    from speechless_amd.engine import HALO
    x, lengths, labels, lab_len, pred_len = _long_form_case()
    case = make_case(b=2, t=64, f=257, seed=15)  # weights / specs of the 257-bin net
    with torch.no_grad():
        tw = tc.to_torch_weights(case["weights"], requires_grad=False)
        ref_probs = tc.forward_probs(case["ospecs"], tw, torch.from_numpy(x))
        ref_losses = tc.per_utterance_ctc(ref_probs, labels, pred_len, lab_len).numpy()
    ref_probs = ref_probs.numpy()
    want_decoded = o.greedy_decode_indices(ref_probs, pred_len)
    for dtype, loss_tol in (("f32", 2e-5), ("bf16", 1e-3)):
        eng = make_engine(case, dtype)
        eng.load_input(x)
        eng.set_labels(labels, np.array(lab_len), np.array(pred_len))
        probs = eng.forward().cpu().numpy()
        losses = eng.ctc().cpu().numpy()
        eng.backward()
        torch.cuda.synchronize()
        err = float(np.abs(losses / ref_losses - 1).max())
        _report("config5_full_length_loss_rel_err_vs_cpu_{}".format(dtype), err)
        assert err < loss_tol, (dtype, losses, ref_losses)
        buf = eng.cur
        assert buf.t_out == 4000 and buf.tt_pad == 4096
        g_out = buf.g[len(eng.plans) - 1].float().cpu().numpy()
        for b, n in enumerate(pred_len):
            assert not g_out[b, HALO + n:].any(), b           # unscored frames: zero gradient (and zero halo / tail)
            assert g_out[b, HALO:HALO + n, :29].any(), b
        assert not g_out[:, :HALO].any() and not g_out[:, :, 29:].any()
        decoded, frame_argmax = eng.greedy_decode(pred_len)
        agree = np.mean([np.mean(frame_argmax[b, :n] == ref_probs[b, :n].argmax(axis=1))
                         for b, n in enumerate(pred_len)])
        _report("config5_full_length_argmax_agreement_{}".format(dtype), float(agree))
        if dtype == "f32":
            assert decoded == want_decoded
            for b, n in enumerate(pred_len):
                assert np.abs(probs[b, :n] - ref_probs[b, :n]).max() < 2e-5
        else:
            assert agree > 0.995
        grads = eng.get_gradients()
        assert all(np.isfinite(dw).all() and np.isfinite(db).all() for dw, db in grads)


# ------------------------------------------------------------------------------------------ bf16 gradients, 1000 frames
def test_bf16_gradients_at_full_length_against_the_cpu_path():
    """What the bf16 path's gradients are worth at configuration-3 size (4 x 1000 frames, labels up to 200), measured
    against the torch-CPU fp32 path, per tensor.  north_star's 1e-3 is met by the loss and by the fp32 path's gradients;
    bf16 STORAGE OF THE ACTIVATIONS cannot meet it for the gradients at random init, whatever the kernels do: a rounding
    step of 2^-9 flips the sign of ~1e-3 of the pre-activations (those within rounding of zero), each flip removes or
    adds one whole element of the back-propagated signal, and sqrt(1e-3) = 3 % per ReLU layer accumulates downwards
    (the float64 oracle run with the same rounding points shows the same figures, with the gradient signal kept in
    fp32 or not: tests/test_oracle.py::test_bf16_storage_noise_is_in_the_activations_not_in_g).  So the bar here is the
    oracle's own bf16 mirror: per layer no worse than 1.5x the mirror's error + 2e-3, and the absolute figures are
    reported (gpurun_out/parity.json) and bounded so that a regression shows."""
    import torch
    from oracle import w2l_torch_cpu as tc
    case = make_case(b=4, t=1000, seed=41)
    rng = np.random.RandomState(41)
    lab_len = [200, 137, 20, 75]
    labels = o.pack_label_batch([list(rng.randint(0, case["k"] - 1, size=n)) for n in lab_len])
    pred_len = [500, 500, 480, 500]
    ref = tc.loss_and_gradients(case["ospecs"], case["weights"], case["x"], labels, pred_len, lab_len)
    errs = {}
    for dtype in ("f32", "bf16"):
        eng = make_engine(case, dtype)
        eng.load_input(case["x"])
        eng.set_labels(labels, np.array(lab_len), np.array(pred_len))
        eng.forward()
        eng.ctc()
        eng.backward()
        torch.cuda.synchronize()
        errs[dtype] = [(rel_l2(dw, rw), rel_l2(db, rb)) for (dw, db), (rw, rb) in zip(eng.get_gradients(), ref["grads"])]
        _report("full_length_gradient_rel_l2_vs_cpu_{}".format(dtype),
                {s.name: [float(a), float(b)] for s, (a, b) in zip(eng.specs, errs[dtype])})
    names = [s.name for s in case["specs"]]
    # Bounds = what was measured (profiles/r02k_parity.json, gpurun_out/parity.json) + 20 %, per tensor, so that a regression
    # of any single layer shows.  fp32 path: the floor of 2.5e-4 on EVERY tensor is the reference's, not the kernels' -- the
    # torch-CPU CTC is an fp32 log-domain lattice (2.5e-4 at 500 frames, HISTORY.md section 3.3), the HIP lattice runs in
    # doubles; on top of it the ReLU flips between two fp32 summation orders (see the config-5 test) in the lowest layers.
    f32_bounds = {"striding_conv": 4.7e-3, "inner_conv_1": 7.5e-4, "inner_conv_2": 5.4e-4, "inner_conv_3": 4.0e-4}
    for name, (ew, eb) in zip(names, errs["f32"]):
        assert ew < f32_bounds.get(name, 3.4e-4) and eb < f32_bounds.get(name, 3.4e-4), ("f32", name, ew, eb)
    # bf16 path (measured: 1.75e-3 at output_conv ... 0.167 at striding_conv), top of the stack downwards
    bounds = {"output_conv": 2.1e-3, "big_conv_2": 4.1e-3, "big_conv_1": 5.9e-3, "inner_conv_7": 6.5e-3,
              "inner_conv_6": 8.6e-3, "inner_conv_5": 1.1e-2, "inner_conv_4": 1.38e-2, "inner_conv_3": 1.65e-2,
              "inner_conv_2": 2.2e-2, "inner_conv_1": 3.15e-2, "striding_conv": 0.2}
    for name, (ew, eb) in zip(names, errs["bf16"]):
        assert ew < bounds[name] and eb < bounds[name], ("bf16", name, ew, eb)
    # errors grow monotonically (within noise) from the output layer down: no single layer is "broken"
    ews = [e for e, _ in errs["bf16"]]
    assert all(ews[i] < 2.5 * ews[i - 1] + 1e-3 for i in range(len(ews) - 1, 0, -1)), ews


def test_bf16_and_fp32_training_trajectories_stay_together():
    """200 Adam(1e-4) steps on one synthetic batch of 32 utterances (128 mel x 200 frames), four runs from the same
    data: the fp32 path; the fp32 path from initial weights perturbed by 1e-6 relative (how chaotic is the optimisation
    itself?); the fp32 path from weights perturbed by 2e-3 relative -- the size of the rounding bf16 storage applies to
    every weight and activation (2^-9); and the bf16 path.
    Measured (tools/trajectory_probe.py): through the well-conditioned first 100 steps (loss 270 -> 62) the bf16 curve
    stays within 7e-4 .. 3e-3 of the fp32 one; after that the optimisation itself is chaotic -- the 1e-6 perturbation of the
    fp32 run has grown to a 10 % loss gap by step 160 and to O(1) by step 180 -- and the bf16 gap follows the envelope
    of a perturbation of its own size.  WHEN a run leaves the common curve is itself chaotic (kernel changes that only
    alter rounding move it by ten steps), so envelopes are compared with 15 steps of slack.  So: (1) bf16 within 5e-3 of
    fp32 at each of the first 100 steps; (2) at every step up to 150 the running maximum of the bf16 gap is at most 4x what
    the 2e-3-perturbed fp32 run reaches within the next 15 steps (+ 2e-3), as long as the latter is below 0.3; (3) all runs
    make progress.  I.e. the gradient noise of bf16 storage is noise of the size the problem already amplifies, not a
    bias."""
    import torch
    b, t, steps = 32, 200, 200
    case = make_case(b=b, t=t, seed=77)
    rng = np.random.RandomState(78)
    lab_len = rng.randint(5, 31, size=b)
    labels = o.pack_label_batch([list(rng.randint(0, 28, size=n)) for n in lab_len])
    pred_len = np.full((b,), t // 2, dtype=np.int32)

    def perturbed(scale, seed):
        prng = np.random.RandomState(seed)
        return [((w * (1 + scale * prng.randn(*w.shape))).astype(np.float32), bb) for w, bb in case["weights"]]

    curves = {}
    for name, dtype, weights in (("f32", "f32", case["weights"]), ("f32_perturbed", "f32", perturbed(1e-6, 5)),
                                 ("f32_perturbed_2e-3", "f32", perturbed(2e-3, 6)), ("bf16", "bf16", case["weights"])):
        eng = make_engine(case, dtype)
        eng.set_weights(weights)
        eng.load_input(case["x"])
        eng.set_labels(labels, lab_len, pred_len)
        means = [eng.train_step_resident().mean() for _ in range(steps)]
        torch.cuda.synchronize()
        curves[name] = np.array([float(m.item()) for m in means])
    a, p, q, c = curves["f32"], curves["f32_perturbed"], curves["f32_perturbed_2e-3"], curves["bf16"]
    gap_b = np.maximum.accumulate(np.abs(c / a - 1))
    gap_p = np.maximum.accumulate(np.abs(p / a - 1))
    gap_q = np.maximum.accumulate(np.abs(q / a - 1))
    marks = (49, 99, 149, 199)
    _report("trajectory_bf16_vs_f32_gap_envelope_at_steps_50_100_150_200", [float(gap_b[i]) for i in marks])
    _report("trajectory_f32_perturbed_1e-6_gap_envelope_at_steps_50_100_150_200", [float(gap_p[i]) for i in marks])
    _report("trajectory_f32_perturbed_2e-3_gap_envelope_at_steps_50_100_150_200", [float(gap_q[i]) for i in marks])
    _report("trajectory_losses_first_100th_last", {k: [float(v[0]), float(v[99]), float(v[-1])] for k, v in curves.items()})
    assert gap_b[99] < 5e-3, gap_b[99]
    assert gap_p[99] < 5e-3 and gap_p[-1] > 0.05, gap_p[marks,]  # the optimisation amplifies 1e-6 to O(0.1 .. 1) by itself
    # compared up to step 150: beyond it every run has left the common curve (loss spikes of O(1) in all four)
    last = 150
    ahead = gap_q[np.minimum(np.arange(last) + 15, steps - 1)]  # what the bf16-sized perturbation reaches 15 steps on
    comparable = ahead < 0.3
    worst = float(np.max(gap_b[:last][comparable] / (4 * ahead[comparable] + 2e-3)))
    _report("trajectory_bf16_gap_over_4x_2e-3_perturbed_envelope_max", worst)
    assert worst <= 1.0, (gap_b[:last][comparable][::10], ahead[comparable][::10])
    for v in curves.values():
        assert v[99] < 0.3 * v[0] and v[:150].min() < 0.2 * v[0]


# ------------------------------------------------------------------------------------------ transfer learning (f1)
SMALL = dict(main_filter_count=20, out_filter_count=40, inner_count=1)


def test_transfer_learning_reindexes_the_output_layer_like_the_reference(tmp_path):
    """Wav2Letter(load_model_from_directory=..., allowed_characters_for_loaded_model=english) for a German net
    (reference net.py:184-269, configuration.py:159-215): all layers copied, the output layer's last axis re-indexed by
    character -- including the reference's `if index` quirk (net.py:254,258): source index 0 ('a') counts as missing and
    gets zeros, like the umlauts that English does not have; the blank maps to the blank."""
    from speechless_amd import Wav2Letter, english_frequent_characters, german_frequent_characters
    en = Wav2Letter(128, english_frequent_characters, seed=4, layer_sizes=SMALL, compute_dtype="f32")
    weights = en.predictive_net.get_weights()
    rng = np.random.RandomState(5)
    weights = [(w, rng.randn(*b.shape).astype(np.float32)) for w, b in weights]  # non-zero biases
    en.predictive_net.set_weights(weights)
    en.predictive_net.save_weights(tmp_path / Wav2Letter.model_file_name(7))
    de = Wav2Letter(128, german_frequent_characters, seed=9, layer_sizes=SMALL, compute_dtype="f32",
                    load_model_from_directory=tmp_path, load_epoch=7,
                    allowed_characters_for_loaded_model=english_frequent_characters)
    got = de.predictive_net.get_weights()
    for i in range(len(got) - 1):
        assert np.array_equal(got[i][0], weights[i][0]) and np.array_equal(got[i][1], weights[i][1]), i
    kernel, bias = got[-1]
    src_kernel, src_bias = weights[-1]
    assert kernel.shape == (1, 40, len(german_frequent_characters) + 1)
    for target, ch in enumerate(german_frequent_characters):
        source = english_frequent_characters.index(ch) if ch in english_frequent_characters else None
        if source:  # the quirk: index 0 is falsy
            assert np.array_equal(kernel[:, :, target], src_kernel[:, :, source]) and bias[target] == src_bias[source]
        else:
            assert ch in "aäöüß" and not kernel[:, :, target].any() and bias[target] == 0, ch
    assert np.array_equal(kernel[:, :, -1], src_kernel[:, :, -1]) and bias[-1] == src_bias[-1]  # blank -> blank
    # and the surgery happened in HBM too: the German net's forward equals the oracle on the surgically built weights
    x = rng.randn(2, 40, 128).astype(np.float32)
    probs = de.prediction_batch(x)
    ospecs = o.layer_specs(128, len(german_frequent_characters) + 1, **SMALL)
    want = o.forward_stack(ospecs, [(w.astype(np.float64), b.astype(np.float64)) for w, b in got],
                           x.astype(np.float64))
    assert np.abs(probs - want).max() < 1e-5


def test_transfer_learning_with_frozen_layers_and_reinitialised_top(tmp_path):
    """frozen_layer_count + reinitialize_trainable_loaded_layers (net.py:179-182,335-339; the `_freeze-8` /
    `-reinitialize` runs of main.py:45-84): only the first `frozen_layer_count` layers are loaded, the rest keep their
    fresh initialisation, and training leaves the frozen ones untouched."""
    from speechless_amd import Wav2Letter, english_frequent_characters, german_frequent_characters
    from speechless_amd.net import Adam
    en = Wav2Letter(128, english_frequent_characters, seed=4, layer_sizes=SMALL)
    en.predictive_net.save_weights(tmp_path / Wav2Letter.model_file_name(3))
    src = en.predictive_net.get_weights()
    fresh = Wav2Letter(128, german_frequent_characters, seed=9, layer_sizes=SMALL).predictive_net.get_weights()
    de = Wav2Letter(128, german_frequent_characters, seed=9, layer_sizes=SMALL, optimizer=Adam(1e-3),
                    load_model_from_directory=tmp_path, load_epoch=3, frozen_layer_count=2,
                    reinitialize_trainable_loaded_layers=True,
                    allowed_characters_for_loaded_model=english_frequent_characters)
    got = de.predictive_net.get_weights()
    for i, ((w, b), (sw, sb), (fw, fb)) in enumerate(zip(got, src, fresh)):
        assert np.array_equal(w, sw if i < 2 else fw) and np.array_equal(b, sb if i < 2 else fb), i
    assert [l.trainable for l in de.predictive_net.layers] == [False, False] + [True] * (len(got) - 2)
    batch = synthetic_examples(4, np.random.RandomState(3))
    for ex in batch:
        ex.label = ex.label.replace("x", "ä")
    before = de.test_and_predict_batch(batch).average_loss
    for _ in range(8):
        de.train_on_batch(batch)
    after = de.predictive_net.get_weights()
    for i in range(2):
        assert np.array_equal(after[i][0], got[i][0]) and np.array_equal(after[i][1], got[i][1])
    assert not np.array_equal(after[2][0], got[2][0])
    assert de.test_and_predict_batch(batch).average_loss < before
    with pytest.raises(ValueError):
        Wav2Letter(128, german_frequent_characters, frozen_layer_count=2)  # net.py:144-145


def test_keras_hdf5_checkpoints_in_class(tmp_path):
    """PredictiveNet.save_weights / load_weights (net.py:209-212, 558-572) on real HDF5, in class, with no h5py in the
    interpreter: the file written is HDF5 in Keras' layout (speechless_amd/h5lite.py; tests/test_h5lite.py has the real
    library read it), it loads back bit for bit, the .npz twin of older versions is still found, and a missing file
    raises."""
    from speechless_amd import Wav2Letter, english_frequent_characters, h5lite
    a = Wav2Letter(128, english_frequent_characters, seed=1, layer_sizes=SMALL)
    b = Wav2Letter(128, english_frequent_characters, seed=2, layer_sizes=SMALL)
    a.predictive_net.save_weights(tmp_path / "weights-epoch1.h5")
    assert sorted(p.name for p in tmp_path.iterdir()) == ["weights-epoch1.h5"]
    assert (tmp_path / "weights-epoch1.h5").read_bytes()[:8] == h5lite.SIGNATURE
    tree = h5lite.read(tmp_path / "weights-epoch1.h5")
    assert tree.attrs["layer_names"] == [layer.name for layer in a.predictive_net.layers]
    assert tree["big_conv_1"].attrs["weight_names"] == ["big_conv_1/kernel:0", "big_conv_1/bias:0"]
    b.predictive_net.load_weights(tmp_path / "weights-epoch1.h5")
    for (w1, b1), (w2, b2) in zip(a.predictive_net.get_weights(), b.predictive_net.get_weights()):
        assert np.array_equal(w1, w2) and np.array_equal(b1, b2)
    # the .npz twin (what rounds 1-2 wrote where h5py was missing) is used when the .h5 asked for does not exist
    a.predictive_net.save_weights(tmp_path / "weights-epoch5.npz")
    c = Wav2Letter(128, english_frequent_characters, seed=3, layer_sizes=SMALL)
    c.predictive_net.load_weights(tmp_path / "weights-epoch5.h5")
    for (w1, _), (w2, _) in zip(a.predictive_net.get_weights(), c.predictive_net.get_weights()):
        assert np.array_equal(w1, w2)
    with pytest.raises(OSError):
        b.predictive_net.load_weights(tmp_path / "weights-epoch2.h5")


def test_keras_written_checkpoint_loads_into_the_engine():
    """A file the REAL HDF5 library wrote in Keras 2.0's save_weights() layout -- Dropout / Lambda groups without weights
    included -- and its model.save() variant (tree under `model_weights`), both committed under tests/golden/
    (make_keras_h5_fixture.py), loaded through PredictiveNet.load_weights into an engine of the fixture's toy topology."""
    from speechless_amd import Wav2Letter
    golden = Path(__file__).resolve().parent / "golden"
    expected = np.load(str(golden / "keras_h5_expected.npz"))
    for name in ("keras_weights_toy.h5", "keras_model_toy.h5"):
        net = Wav2Letter(4, "abcd", seed=0, layer_sizes=TOY_SIZES, compute_dtype="f32")
        net.predictive_net.load_weights(golden / name)
        for layer, (w, b) in zip(net.predictive_net.layers, net.predictive_net.get_weights()):
            assert np.array_equal(w, expected[layer.name + "/kernel"]) and np.array_equal(b, expected[layer.name + "/bias"])
    with pytest.raises(ValueError, match="holds no weights for layer"):
        Wav2Letter(4, "abcd", seed=0, layer_sizes=dict(TOY_SIZES, inner_count=3),
                   compute_dtype="f32").predictive_net.load_weights(golden / "keras_weights_toy.h5")


# ------------------------------------------------------------------------------------------ BatchNorm fold (a12)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("layer", [1, 8])
def test_folded_batchnorm_runs_in_the_fused_conv_bias_relu_epilogue(dtype, layer):
    """north_star's 'fused BatchNorm+ReLU' (absent from the reference) on the GPU: an inference-mode
    BatchNorm behind layer `layer` is folded into that layer's kernel and bias (speechless_amd/fold.py) and the ONE
    sl_conv1d_nt launch with the BIAS_RELU epilogue must then equal relu(BatchNorm(conv(x) + b)) computed by the oracle
    in float64 with the un-folded parameters -- no normalisation pass over HBM."""
    import torch
    from speechless_amd import _lib
    from speechless_amd.engine import HALO
    from speechless_amd.fold import fold_batchnorm_into_conv
    case = make_case(b=3, t=150, seed=30)
    eng = make_engine(case, dtype)
    buf = eng.load_input(case["x"])
    p = eng.plans[layer]
    s = p.spec
    rng = np.random.RandomState(60 + layer)
    t_out = buf.t_out
    x_in = np.maximum(o.round_to_bf16(rng.randn(3, t_out, s.cin).astype(np.float32)), 0)
    xt = torch.zeros_like(buf.y[layer - 1])
    xt[:, HALO:HALO + t_out, :s.cin] = torch.tensor(x_in).to(eng.torch_dtype)
    buf.y[layer - 1].copy_(xt)
    w, b = case["weights"][layer]
    gamma, beta = rng.uniform(0.5, 1.5, s.cout), rng.randn(s.cout) * 0.1
    mean, var, eps = rng.randn(s.cout) * 0.05, rng.uniform(0.2, 2.0, s.cout), 1e-3
    z = o.conv1d_preactivation(x_in.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 1)
    want = np.maximum(gamma * (z - mean) / np.sqrt(var + eps) + beta, 0)
    w2, b2 = fold_batchnorm_into_conv(w, b, gamma, beta, mean, var, eps)
    weights = list(case["weights"])
    weights[layer] = (w2, b2)
    eng.set_weights(weights)
    eng.repack_weights()
    st = torch.cuda.current_stream().cuda_stream
    eng.lib.call("sl_conv1d_nt", buf.y[layer - 1].data_ptr(), eng.w_fwd[layer].data_ptr(),
                 eng.layer_param_views(eng.params, p)[1].data_ptr(), None, buf.y[layer].data_ptr(),
                 ctypes.byref(buf.fwd_geom[layer]), _lib.EPI_BIAS_RELU, eng.dtype_code, 0, 0, buf.nt_ws.data_ptr(),
                 buf.nt_ws.numel(), st)
    torch.cuda.synchronize()
    got, _ = layer_activation(eng, buf, layer)
    err = rel_l2(got, want)
    _report("folded_batchnorm_rel_l2_{}_layer{}".format(dtype, layer), err)
    assert err < (1e-5 if dtype == "f32" else 6e-3), err  # bf16: folded weights and the output are rounded to bf16


# ------------------------------------------------------------------------------------------ optimizer state (f3)
def test_optimizer_state_checkpoint_resumes_bitwise(tmp_path):
    """Extension behind a flag (the reference saves weights only, net.py:564-572): k steps, save weights + Adam
    moments + step count, load into a fresh net, k more steps == 2k uninterrupted steps, bit for bit; without the
    optimizer state the resumed run differs (Adam restarts its moments)."""
    from speechless_amd import Wav2Letter, english_frequent_characters
    from speechless_amd.net import Adam
    batch = synthetic_examples(4, np.random.RandomState(21))

    def net(**kw):
        return Wav2Letter(128, english_frequent_characters, seed=6, layer_sizes=SMALL, optimizer=Adam(1e-3),
                          dropout=0.1, **kw)
    straight = net()
    for _ in range(6):
        straight.train_on_batch(batch)
    first = net()
    for _ in range(3):
        first.train_on_batch(batch)
    first.predictive_net.save_weights(tmp_path / Wav2Letter.model_file_name(1))
    first.save_optimizer_state(tmp_path, 1)
    resumed = net(load_model_from_directory=tmp_path, load_epoch=1, load_optimizer_state=True)
    cold = net(load_model_from_directory=tmp_path, load_epoch=1)
    for _ in range(3):
        resumed.train_on_batch(batch)
        cold.train_on_batch(batch)
    for (w1, b1), (w2, b2), (w3, b3) in zip(straight.predictive_net.get_weights(), resumed.predictive_net.get_weights(),
                                            cold.predictive_net.get_weights()):
        assert np.array_equal(w1, w2) and np.array_equal(b1, b2)
    assert any(not np.array_equal(w1, w3) for (w1, _), (w3, _) in zip(straight.predictive_net.get_weights(),
                                                                     cold.predictive_net.get_weights()))
    # train(save_optimizer_state=True) writes the file next to every checkpoint
    straight.train([batch] * 8, preview_labeled_spectrogram_batch=batch[:2], tensor_board_log_directory=None,
                   net_directory=tmp_path / "run", batches_per_epoch=4, save_optimizer_state=True)
    assert sorted(p.name for p in (tmp_path / "run").iterdir())[0].startswith("weights-epoch1")
    assert (tmp_path / "run" / Wav2Letter.optimizer_state_file_name(1)).exists()


# ------------------------------------------------------------------------------------------ dropout
def _dropout_keep(seed, n, rate):
    """numpy mirror of dropout_bits (csrc/misc.hip): splitmix64 finaliser of seed + golden * (index + 1)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * idx
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(32)).astype(np.uint32) >= np.uint32(int(rate * 4294967296.0))


@pytest.mark.parametrize("activation", ["elu", "relu"])
def test_dropout_training_step_with_recomputed_masks(activation):
    """Dropout in front of the first n-3 layers with ELU (and ReLU) hidden layers, fp32 path: the keep decisions are a
    pure function of (seed, element index), so the test recomputes every mask on the host, hands them to the oracle as
    explicit multipliers and expects loss and all gradients to agree as tightly as without dropout.  (Behind an ELU a
    stored zero does not identify a dropped element; the backward pass recomputes the decisions too.)"""
    import torch
    from speechless_amd.engine import Engine, HALO, wav2letter_layer_specs
    case = make_case(b=3, t=96, seed=9)
    rate = 0.25
    specs = wav2letter_layer_specs(128, 29, activation=activation)
    ospecs = o.layer_specs(128, 29, activation=activation)
    eng = Engine(specs, 29, dtype="f32")
    eng.set_weights(case["weights"])
    eng.dropout_rate, eng.dropout_seed = rate, 5
    eng.load_input(case["x"])
    eng.set_labels(case["labels"], np.array(case["label_lengths"]), np.array(case["prediction_lengths"]))
    eng.forward(training=True)
    loss = eng.ctc().cpu().numpy().copy()
    eng.backward()
    torch.cuda.synchronize()
    buf = eng.cur
    n = len(eng.plans)
    seed0 = (5 * 1000003 + 1) * 64
    b, t_in, f = case["x"].shape
    p0 = eng.plans[0]
    keep = _dropout_keep(seed0, buf.x0.numel(), rate).reshape(tuple(buf.x0.shape))
    scales = [keep[:, p0.pad_left:p0.pad_left + t_in, :f] / (1 - rate)] + [None] * (n - 1)
    for i in range(1, n - 3):
        y = buf.y[i - 1]
        keep = _dropout_keep(seed0 + i, y.numel(), rate).reshape(tuple(y.shape))
        scales[i] = keep[:, HALO:HALO + buf.t_out, :specs[i].cin] / (1 - rate)
    ref = o.loss_and_gradients(ospecs, weights64(case), case["x"].astype(np.float64), case["labels"],
                               case["prediction_lengths"], case["label_lengths"], input_scales=scales)
    assert np.allclose(loss, ref["losses"], rtol=2e-5), (loss, ref["losses"])
    errs = [max(rel_l2(dw, rw), rel_l2(db, rb)) for (dw, db), (rw, rb) in zip(eng.get_gradients(), ref["grads"])]
    # fp32 against float64: a pre-activation within fp32 rounding of zero may take the other ReLU branch (one mask
    # element of ~36 000 here = 5e-3 of the signal from that layer DOWN, HISTORY.md section 1 "ReLU mask flips"); so the layers
    # above the first such flip agree to 2e-4 and the ones below it, a prefix of the stack, to 1e-2
    loose = [i for i, e in enumerate(errs) if e >= 2e-4]
    assert max(errs) < 1e-2 and loose == list(range(len(loose))) and errs[-1] < 2e-4, errs


def test_dropout_without_a_seed_draws_one():
    """The reference signature has no seed: Wav2Letter(..., dropout=0.1) must train."""
    from speechless_amd import Wav2Letter, english_frequent_characters
    batch = synthetic_examples(4, np.random.RandomState(17))
    net = Wav2Letter(128, english_frequent_characters, dropout=0.1, layer_sizes=SMALL)
    assert isinstance(net.engine.dropout_seed, int)
    assert np.isfinite(net.train_on_batch(batch))
    elu = Wav2Letter(128, english_frequent_characters, dropout=0.1, activation="elu", layer_sizes=SMALL)
    assert np.isfinite(elu.train_on_batch(batch))


# ------------------------------------------------------------------------------------------ variable-length batches
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_batches_of_different_lengths_share_one_set_of_buffers(dtype):
    """The reference's generator pads every batch to its own longest member (corpus.py:224-226), so the frame count
    changes with nearly every step.  Batches whose output frames round up to the same multiple of 256 run in the same
    HBM buffers; what a longer batch left behind beyond a shorter one's valid rows must never be seen: loss and
    gradients of every batch equal, bit for bit, those of a fresh engine that has only ever seen that batch."""
    import torch
    lengths = [300, 262, 131, 300, 508, 77]
    cases = {t: make_case(b=3, t=t, seed=50 + t) for t in set(lengths)}
    for case in cases.values():
        case["weights"] = cases[300]["weights"]  # (make_case draws the biases from its seed)
    shared = make_engine(cases[300], dtype)
    for t in lengths:
        case = cases[t]
        losses, _ = run_loss_and_grads(shared, case)
        got = shared.grads.clone()
        assert len(shared._buffers) == 1 and shared.cur.t_in == t
        fresh = make_engine(case, dtype)
        want_losses, _ = run_loss_and_grads(fresh, case)
        assert np.array_equal(losses, want_losses), t
        assert torch.equal(got, fresh.grads), t
        dec_a, _ = shared.greedy_decode(case["prediction_lengths"])
        dec_b, _ = fresh.greedy_decode(case["prediction_lengths"])
        assert dec_a == dec_b
    big = make_case(b=3, t=600, seed=1)  # T' = 300 -> the next buffer size
    big["weights"] = cases[300]["weights"]
    run_loss_and_grads(shared, big)
    assert len(shared._buffers) == 2


# ------------------------------------------------------------------------------------------ exact-fp32 MFMA path
def test_fp32_mfma_kernels_against_the_valu_kernels_and_float64():
    """The fp32 path's two GEMMs on v_mfma_f32_32x32x2_f32 (default) against the plain VALU FMA kernels of round 1
    (cfg 1) and against float64, through a whole step: both are exact-fp32 fmaf chains in a different order, so they
    agree to fp32 round-off."""
    from speechless_amd.engine import HALO
    case = make_case(b=3, t=150, seed=12)
    engines = {}
    for name, cfg in (("mfma", 0), ("valu", 1)):
        eng = make_engine(case, "f32")
        for spec in eng.specs:
            for kind in ("fwd", "dgrad", "wgrad"):
                eng.nt_cfg[(kind, spec.name)] = cfg
        losses, grads = run_loss_and_grads(eng, case)
        engines[name] = (eng, losses, grads)
    ref = o.loss_and_gradients(case["ospecs"], weights64(case), case["x"].astype(np.float64), case["labels"],
                               case["prediction_lengths"], case["label_lengths"])
    (ea, la, ga), (eb, lb, gb) = engines["mfma"], engines["valu"]
    np.testing.assert_allclose(la, ref["losses"], rtol=1e-5)
    np.testing.assert_allclose(la, lb, rtol=2e-6)
    assert float((ea.cur.probs - eb.cur.probs).abs().max()) < 2e-6
    for i, ((dw, db), (vw, vb), (rw, rb)) in enumerate(zip(ga, gb, ref["grads"])):
        assert rel_l2(dw, rw) < 2e-4 and rel_l2(db, rb) < 2e-4, (i, rel_l2(dw, rw))
        assert rel_l2(dw, vw) < 2e-4, (i, rel_l2(dw, vw))
    for i in range(len(ea.plans) - 1):
        ya, _ = layer_activation(ea, ea.cur, i)
        yb, _ = layer_activation(eb, eb.cur, i)
        assert np.abs(ya - yb).max() <= 2e-5 * max(np.abs(yb).max(), 1e-30), i
    # layout invariants of the MFMA kernels' masked stores: halo rows, rows beyond T' and padded channels stay zero
    # (but for the ones channel of the activations, Engine.ones_channel)
    for tensors in (ea.cur.y, ea.cur.g[:-1]):
        for i, t in enumerate(tensors):
            raw = t.float().cpu().numpy()
            c = ea.specs[i].cout
            assert not raw[:, :HALO].any() and not raw[:, HALO + ea.cur.t_out:].any(), i
            padding = raw[:, :, c:]
            if tensors is ea.cur.y and ea.ones_channel and i < len(ea.plans) - 1 and padding.shape[2]:
                assert (padding[:, HALO:HALO + ea.cur.t_out, -1] == 1).all(), i
                padding = padding[:, :, :-1]
            assert not padding.any(), i


def test_config2_greedy_decode_bit_exact_at_batch_32():
    """BASELINE config 2 at its stated size on the path bench.py --config 2 reports: random-init Wav2Letter forward on
    32 x 128-mel x 1000 frames, fp32 storage, exact-fp32 MFMA contraction; greedy-decoded label indices (and every
    frame's argmax) bit-exact against the torch-CPU fp32 path.  The bf16 path's disagreements are counted, not hidden."""
    import torch
    from oracle import w2l_torch_cpu as tc
    case = make_case(b=32, t=1000, seed=2)
    pred_len = [500] * 32
    with torch.no_grad():
        ref_probs = tc.forward_probs(case["ospecs"], tc.to_torch_weights(case["weights"], requires_grad=False),
                                     torch.from_numpy(case["x"])).numpy()
    want = o.greedy_decode_indices(ref_probs, pred_len)
    argmax, margin = o.frame_argmax_and_margin(ref_probs)
    _report("config2_b32_min_top1_top2_margin", float(margin.min()))
    eng = make_engine(case, "f32")
    probs = eng.forward(case["x"]).cpu().numpy()
    decoded, frame_argmax = eng.greedy_decode(pred_len)
    assert np.abs(probs - ref_probs).max() < 2e-5
    assert np.array_equal(frame_argmax, argmax)
    assert decoded == want
    eng16 = make_engine(case, "bf16")
    eng16.forward(case["x"])
    decoded16, frame_argmax16 = eng16.greedy_decode(pred_len)
    _report("config2_b32_bf16_mismatching_frames_of_16000", int((frame_argmax16 != argmax).sum()))
    _report("config2_b32_bf16_mismatching_sequences_of_32", int(sum(a != b for a, b in zip(decoded16, want))))
    assert (frame_argmax16 != argmax).mean() < 5e-2  # measured 1.7 % of the frames: the softmax is nearly flat


# ------------------------------------------------------------------------------------------ CTC lattice variants
def _ctc_variant(hip_lib, variant):
    hip_lib.call("sl_ctc_select", variant)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
def test_ctc_lattice_variants_against_the_oracle(hip_lib, variant):
    """sl_ctc_loss_grad's lattices -- probability domain, one wave per utterance and direction, in doubles with an exponent
    per lane and 16 frames (variants 2, 3, 4 and the default) or in floats with one per 8 frames (5, 6, 7) -- and log domain (variant 1; also the repair pass that variants 3 and 7 force for every utterance) -- on the
    edge cases of round 1 (repeats, empty label, input_len < T', no valid alignment), on labels of 200 graphemes over 500
    frames and on TensorFlow's known answers: same tolerances for all of them."""
    import json
    from test_gpu_parity import run_ctc_kernel
    try:
        _ctc_variant(hip_lib, variant)
        rng = np.random.RandomState(9)
        k, t = 7, 40
        labels_list = [[0, 1, 2, 3], [4, 4, 4, 4, 4], [], [0, 5, 0, 5, 0, 5, 0], list(rng.randint(0, 6, size=19)),
                       list(rng.randint(0, 6, size=30))]
        input_len = [40, 40, 12, 13, 40, 25]  # last: 30 labels in 25 frames -> no valid alignment
        logits = (rng.randn(len(labels_list), t, k) * 2).astype(np.float32)
        labels = o.pack_label_batch([l if l else [-1] for l in labels_list])
        lab_len = [len(l) for l in labels_list]
        probs, loss, dl = run_ctc_kernel(hip_lib, logits, labels, lab_len, input_len)
        ref_p = o.softmax(logits.astype(np.float64))
        ref_loss, ref_dp = o.ctc_batch_cost(ref_p, labels, input_len, lab_len)
        ref_dl = o.softmax_backward(ref_p, ref_dp)
        assert np.isinf(ref_loss[5]) and np.isinf(loss[5])
        np.testing.assert_allclose(loss[:5], ref_loss[:5], rtol=1e-5)
        for i in range(len(labels_list)):
            assert np.abs(dl[i] - ref_dl[i]).max() < 1e-4, (i, np.abs(dl[i] - ref_dl[i]).max())
            assert not dl[i, input_len[i]:].any()
        # long labels, full length
        rng = np.random.RandomState(10)
        k, t, b = 29, 500, 4
        lab_len = [200, 137, 1, 60]
        labels = o.pack_label_batch([list(rng.randint(0, 28, size=n)) for n in lab_len])
        logits = rng.randn(b, t, k).astype(np.float32)
        input_len = [500, 480, 500, 333]
        _, loss, dl = run_ctc_kernel(hip_lib, logits, labels, lab_len, input_len)
        ref_p = o.softmax(logits.astype(np.float64))
        ref_loss, ref_dp = o.ctc_batch_cost(ref_p, labels, input_len, lab_len)
        np.testing.assert_allclose(loss, ref_loss, rtol=1e-5)
        err = rel_l2(dl, o.softmax_backward(ref_p, ref_dp))
        _report("ctc_long_labels_gradient_rel_l2_variant{}".format(variant), err)
        assert err < 1e-3
        # TensorFlow's ctc_loss_op_test.py::testBasic (eps = 0: the op's own arithmetic): loss and gradient
        cases = json.loads((GOLDEN / "tf_known_answers.json").read_text())["ctc_loss"]
        lg = np.stack([np.log(np.array(c["probs"], dtype=np.float64)) for c in cases]).astype(np.float32)
        lab = -np.ones((len(cases), 5), dtype=np.int32)
        for i, c in enumerate(cases):
            lab[i, :len(c["labels"])] = c["labels"]
        _, loss, dl = run_ctc_kernel(hip_lib, lg, lab, [len(c["labels"]) for c in cases], [5] * len(cases), eps=0.0)
        for i, c in enumerate(cases):
            assert abs(loss[i] - c["loss"]) < 1e-5, (i, loss[i])
            assert np.abs(dl[i] - np.array(c["grad_logits"])).max() < 3e-6, i
    finally:
        _ctc_variant(hip_lib, 0)


def test_ctc_probability_domain_lattice_in_the_blank_collapse_regime(hip_lib):
    """What a single exponent per lattice ROW could not hold (the first version of the wave lattice reported Z = 0 here):
    140 x 'a' over 300 frames whose first half predicts nothing but blank (p_a at the 1e-8 floor), so that every alignment
    pays ~65 floor emissions (2^-1700) before frame 150 while the all-blank prefix costs nothing; and, more generally, the
    regime every CTC training run passes through early on -- p(blank) ~ 1, every label at the floor -- where the states
    of one row span thousands of binades.  With one exponent per LANE (8 states) the probability-domain lattice handles
    both without the repair pass (variant 2 = no repair launches), to fp32 round-off of the float64 oracle."""
    from test_gpu_parity import run_ctc_kernel
    k, t, n = 3, 300, 140
    logits = np.zeros((2, t, k), dtype=np.float32)
    logits[:, :150, 2] = 40.0
    logits[:, 150::2, 0] = 6.0
    logits[:, 151::2, 2] = 6.0
    labels = o.pack_label_batch([[0] * n, [0, 1, 0]])
    lab_len, input_len = [n, 3], [t, t]
    ref_p = o.softmax(logits.astype(np.float64))
    ref_loss, ref_dp = o.ctc_batch_cost(ref_p, labels, input_len, lab_len)
    assert np.isfinite(ref_loss).all() and ref_loss[0] > 1000
    rng = np.random.RandomState(4)
    k2, t2, n2 = 29, 1500, 120
    logits2 = rng.randn(1, t2, k2).astype(np.float32)
    logits2[:, :, k2 - 1] += 25.0  # blank collapse
    labels2 = o.pack_label_batch([list(rng.randint(0, 28, size=n2))])
    ref_p2 = o.softmax(logits2.astype(np.float64))
    ref_loss2, ref_dp2 = o.ctc_batch_cost(ref_p2, labels2, [t2], [n2])
    ref_dl2 = o.softmax_backward(ref_p2, ref_dp2)
    try:
        for variant in (2, 4):
            _ctc_variant(hip_lib, variant)
            _, loss, dl = run_ctc_kernel(hip_lib, logits, labels, lab_len, input_len)
            np.testing.assert_allclose(loss, ref_loss, rtol=2e-6)
            assert np.abs(dl - o.softmax_backward(ref_p, ref_dp)).max() < 2e-5
            _, loss2, dl2 = run_ctc_kernel(hip_lib, logits2, labels2, [n2], [t2])
            np.testing.assert_allclose(loss2, ref_loss2, rtol=2e-6)
            assert rel_l2(dl2, ref_dl2) < 1e-4, rel_l2(dl2, ref_dl2)
        # float lattice + repair pass: right to the tolerance of its sum-of-posteriors check (4e-3 of Z = 4e-3 absolute in
        # the loss) whatever the float lattice managed by itself
        _ctc_variant(hip_lib, 5)
        _, loss, dl = run_ctc_kernel(hip_lib, logits, labels, lab_len, input_len)
        assert np.abs(loss - ref_loss).max() < 8e-3, loss - ref_loss
        assert np.abs(dl - o.softmax_backward(ref_p, ref_dp)).max() < 5e-3
        _, loss2, dl2 = run_ctc_kernel(hip_lib, logits2, labels2, [n2], [t2])
        assert np.abs(loss2 - ref_loss2).max() < 8e-3, loss2 - ref_loss2
        assert rel_l2(dl2, ref_dl2) < 1e-2, rel_l2(dl2, ref_dl2)
        _ctc_variant(hip_lib, 6)  # float lattice alone: reported for the record (does it need the repair pass here?)
        _, loss, dl = run_ctc_kernel(hip_lib, logits, labels, lab_len, input_len)
        _, loss2, dl2 = run_ctc_kernel(hip_lib, logits2, labels2, [n2], [t2])
        _report("ctc_float_lattice_alone_in_the_collapse_regime", {
            "loss_rel_err": [float(abs(loss[0] - ref_loss[0]) / ref_loss[0]), float(abs(loss2[0] - ref_loss2[0]) / ref_loss2[0])],
            "gradient_rel_l2": [float(rel_l2(dl, o.softmax_backward(ref_p, ref_dp))), float(rel_l2(dl2, ref_dl2))]})
    finally:
        _ctc_variant(hip_lib, 0)


# ------------------------------------------------------------------------------------------ launch lists
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_recorded_launch_lists_train_exactly_like_the_eager_path(dtype):
    """Engine.use_launch_lists (default on): the C-ABI calls and stream hand-overs of forward and backward are recorded
    once per buffer set and replayed with their arguments already marshalled.  Several optimisation steps over batches
    of different lengths (one buffer set, geometries re-targeted in place) and a change of weights in between must give
    bit-identical weights and losses with and without them."""
    import torch
    lengths = [300, 262, 300, 131, 262, 508]
    cases = {t: make_case(b=3, t=t, seed=60 + t) for t in set(lengths)}
    results = []
    for use_lists in (True, False):
        eng = make_engine(cases[300], dtype, lr=1e-3)
        eng.use_launch_lists = use_lists
        losses = []
        for step, t in enumerate(lengths):
            case = cases[t]
            loss = eng.train_step(case["x"], case["labels"], np.array(case["label_lengths"]),
                                  np.array(case["prediction_lengths"]))
            losses.append(loss.cpu().numpy().copy())
            if step == 2:  # weights replaced from outside: the operand copies are repacked before the next replay
                eng.set_weights([(w * 0.5, b) for w, b in eng.get_weights()])
        torch.cuda.synchronize()
        if use_lists:
            assert eng.cur.launch_lists, "nothing was recorded"
        results.append((losses, eng.params.clone(), eng.forward(cases[262]["x"]).clone()))
    for a, b in zip(results[0][0], results[1][0]):
        assert np.array_equal(a, b)
    assert torch.equal(results[0][1], results[1][1]) and torch.equal(results[0][2], results[1][2])


# ------------------------------------------------------------------------------------------ fused run of inner layers
def test_fused_run_of_inner_layers_against_the_single_launches():
    """sl_conv1d_chain (inner_conv_1..7 in one launch, activations handed from layer to layer through LDS, halo
    recomputed per 64-frame tile) against the seven sl_conv1d_nt launches it replaces, forward and input-gradient
    direction, on batches whose frame counts exercise partial tiles, a single tile and the utterance boundaries: every
    stored activation / gradient agrees to bf16 round-off of the same fp32 sums in another order, the layout invariants
    hold, and loss / gradients stay within the bf16 path's tolerances against the oracle's bf16 mirror."""
    import torch
    from speechless_amd.engine import HALO
    for t in (300, 77, 1000):
        case = make_case(b=3, t=t, seed=70 + t)
        res = {}
        for chain in (True, False):
            eng = make_engine(case, "bf16")
            eng.use_chain = chain
            losses, grads = run_loss_and_grads(eng, case)
            tags = [op[3] for ops in eng.cur.launch_lists.values() for op in ops if op[0] == 0]
            assert (tags.count("sl_conv1d_chain") == 2) == chain and ("sl_conv1d_chain" in tags) == chain
            res[chain] = (eng, losses, grads)
        (ea, la, ga), (eb, lb, gb) = res[True], res[False]
        t_out = ea.cur.t_out
        for i in range(len(ea.plans) - 1):
            ya, yb = ea.cur.y[i].float().cpu().numpy(), eb.cur.y[i].float().cpu().numpy()
            c = ea.specs[i].cout
            assert not ya[:, :HALO].any() and not ya[:, HALO + t_out:].any() and not ya[:, :, c:-1].any(), i
            assert (ya[:, HALO:HALO + t_out, -1] == 1).all() and (yb[:, HALO:HALO + t_out, -1] == 1).all(), i  # ones channel
            scale = max(np.abs(yb).max(), 1e-30)
            assert np.abs(ya - yb).max() <= 2.0 ** -6 * scale, (t, "y", i, np.abs(ya - yb).max() / scale)
            assert rel_l2(ya, yb) < 3e-3, (t, "y", i, rel_l2(ya, yb))
        for i in range(len(ea.plans)):
            ga_i, gb_i = ea.cur.g[i].float().cpu().numpy(), eb.cur.g[i].float().cpu().numpy()
            assert not ga_i[:, :HALO].any() and not ga_i[:, HALO + t_out:].any(), i
            assert rel_l2(ga_i, gb_i) < 3e-2, (t, "g", i, rel_l2(ga_i, gb_i))  # (ReLU-mask flips between the two orders)
        np.testing.assert_allclose(la, lb, rtol=2e-3)
        if t == 300:
            mirror = o.loss_and_gradients(case["ospecs"], weights64(case), case["x"].astype(np.float64), case["labels"],
                                          case["prediction_lengths"], case["label_lengths"], bf16_mirror=True)
            np.testing.assert_allclose(la, mirror["losses"], rtol=1e-3)
            for name, i in (("output_conv", 10), ("big_conv_2", 9), ("big_conv_1", 8)):
                assert rel_l2(ga[i][0], mirror["grads"][i][0]) < 2e-2, name


# ------------------------------------------------------------------------------------------ bias gradients out of the wgrad GEMM
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("dropout", [None, 0.3])
def test_bias_gradients_from_the_ones_channel(dropout, dtype):
    """Engine.ones_channel: the last padded output channel of every hidden layer is the constant 1, so the next layer's
    weight-gradient GEMM leaves that layer's bias gradient in row cin_pad - 1 of dW (sl_bias_grad_from_wgrad) and the ten
    sl_bias_grad passes over g are not launched.  Same bias gradients as those passes (fp32 sums of the same bf16 values in
    another order) and as the oracle's bf16 mirror; the padded weights stay exactly zero and the ones stay ones through
    optimisation steps -- also when dropout hits the ones (then the passes are used and the row is only zeroed)."""
    import torch
    case = make_case(b=3, t=300, seed=91)
    res = {}
    for ones in (True, False):
        eng = make_engine(case, dtype, lr=1e-3)
        assert eng.ones_channel
        if not ones:
            eng.ones_channel = False
            eng.set_weights(eng.get_weights())  # (rebuilds the parameter buffer without the ones biases)
        if dropout:
            eng.dropout_rate, eng.dropout_seed = dropout, 5
        eng.load_input(case["x"])
        eng.set_labels(case["labels"], case["label_lengths"], case["prediction_lengths"])
        eng.forward(training=True)
        losses = eng.ctc().cpu().numpy()
        eng.backward()
        torch.cuda.synchronize()
        grads = eng.get_gradients()
        tags = [op[3] for ops in eng.cur.launch_lists.values() for op in ops if op[0] == 0] if not dropout else []
        if not dropout:
            assert tags.count("sl_bias_grad") == (1 if ones else len(eng.plans))
            assert tags.count("sl_bias_grad_from_wgrad") == (1 if ones else 0)
        res[ones] = (eng, losses, grads)
    (ea, la, ga), (eb, lb, gb) = res[True], res[False]
    np.testing.assert_allclose(la, lb, rtol=1e-6)
    for i in range(len(ea.plans)):
        assert rel_l2(ga[i][0], gb[i][0]) < 1e-6, i      # weight gradients: the same kernels on the same data
        assert rel_l2(ga[i][1], gb[i][1]) < 1e-5, (i, rel_l2(ga[i][1], gb[i][1]))  # bias gradients: GEMM row vs pass over g
    # optimisation steps: padded weights stay zero, the ones stay ones, both engines stay together
    for eng in (ea, eb):
        for _ in range(3):
            eng.train_step(case["x"], case["labels"], np.array(case["label_lengths"]), np.array(case["prediction_lengths"]))
    for p in ea.plans:
        wv, bv = ea.layer_param_views(ea.params, p)
        wv, bv = wv.cpu().numpy(), bv.cpu().numpy()
        assert not wv[:, p.spec.cin:, :].any() and not wv[:, :, p.spec.cout:].any(), p.spec.name
        assert not bv[p.spec.cout:p.cout_pad - 1].any() and bv[p.cout_pad - 1] == (1.0 if ea._has_ones_output(p) else 0.0)
    if not dropout:
        for (wa, ba), (wb, bb) in zip(ea.get_weights(), eb.get_weights()):
            assert rel_l2(wa, wb) < 1e-4 and np.abs(ba - bb).max() < 1e-4 * max(np.abs(bb).max(), 1e-6) + 1e-6


# ------------------------------------------------------------------------------------------ the round's fast paths over odd shapes
@pytest.mark.parametrize("b,t,frozen", [(1, 33, 0), (2, 127, 0), (5, 129, 0), (3, 255, 4), (7, 641, 0), (4, 1031, 6)])
def test_fast_paths_against_the_plain_launches_over_odd_shapes(b, t, frozen):
    """Fused inner-layer launches + bias gradients out of the weight-gradient GEMM + wave CTC lattice (the defaults) against
    single launches + sl_bias_grad passes + log-domain lattice on batches whose sizes hit the corners of the tilings: one
    utterance, fewer frames than a 64-frame tile, one frame more than a tile / than a 256-row time tile, ragged
    prediction lengths; also with the first 4 / 6 layers frozen (the fused input-gradient launch then covers only the
    upper part of the run).  Same loss; gradients as close as two summation orders of the same bf16 data allow."""
    import torch
    from speechless_amd import _lib
    case = make_case(b=b, t=t, seed=200 + t)
    res = {}
    try:
        for fast in (True, False):
            eng = make_engine(case, "bf16", frozen_layer_count=frozen)
            if not fast:
                eng.use_chain = False
                eng.ones_channel = False
                eng.set_weights(eng.get_weights())
            _lib.lib().call("sl_ctc_select", 0 if fast else 1)
            losses, grads = run_loss_and_grads(eng, case)
            res[fast] = (losses, grads, eng)
    finally:
        _lib.lib().call("sl_ctc_select", 0)
    (la, ga, ea), (lb, gb, _) = res[True], res[False]
    tags = [op[3] for ops in ea.cur.launch_lists.values() for op in ops if op[0] == 0]
    assert "sl_conv1d_chain" in tags and "sl_bias_grad_from_wgrad" in tags
    np.testing.assert_allclose(la, lb, rtol=2e-3)  # (bf16 activations through two different accumulation orders)
    for i in range(frozen, len(ga)):
        assert rel_l2(ga[i][0], gb[i][0]) < 5e-2, (i, rel_l2(ga[i][0], gb[i][0]))  # ReLU-mask flips between the orders
        assert rel_l2(ga[i][1], gb[i][1]) < 5e-2, (i, rel_l2(ga[i][1], gb[i][1]))
    for i in (8, 9, 10):  # above the fused run the two paths see the same activations up to those flips' echo
        assert rel_l2(ga[i][0], gb[i][0]) < 2e-2, i


def test_ctc_falls_back_to_the_log_domain_lattice_where_the_wave_lattice_does_not_fit(hip_lib):
    """The wave lattice holds 2 * 255 + 1 states and 63 classes; beyond that sl_ctc_loss_grad must pick the log-domain
    lattice by itself (it was the default before, so long labels used to be covered implicitly): labels of 300 graphemes
    over 700 frames, and 64 classes, against the float64 oracle."""
    from test_gpu_parity import run_ctc_kernel
    rng = np.random.RandomState(17)
    for k, n, t in ((29, 300, 700), (64, 40, 120)):
        lab_len = [n, n // 3]
        labels = o.pack_label_batch([list(rng.randint(0, k - 1, size=m)) for m in lab_len])
        logits = rng.randn(2, t, k).astype(np.float32)
        input_len = [t, t - 7]
        _, loss, dl = run_ctc_kernel(hip_lib, logits, labels, lab_len, input_len)
        ref_p = o.softmax(logits.astype(np.float64))
        ref_loss, ref_dp = o.ctc_batch_cost(ref_p, labels, input_len, lab_len)
        np.testing.assert_allclose(loss, ref_loss, rtol=2e-5)
        assert rel_l2(dl, o.softmax_backward(ref_p, ref_dp)) < 2e-3, (k, n)
