"""CPU tests of the oracle (oracle/w2l_oracle.py): golden vectors, independent cross-checks, finite differences."""
from pathlib import Path

import numpy as np
import pytest

from oracle import w2l_oracle as o
from oracle import w2l_torch_cpu as tc

GOLDEN = Path(__file__).resolve().parent / "golden"


def toy_specs():
    return o.layer_specs(4, 5, main_filter_count=6, out_filter_count=8, striding_kernel=6, inner_kernel=3,
                         big_kernel=4, inner_count=2)


def rebuild_case(name):
    g = np.load(str(GOLDEN / "stack_golden.npz"))
    specs = toy_specs() if name == "toy" else o.layer_specs(128, 29)
    weights = o.glorot_uniform_weights(specs, seed=2, dtype=np.float64)
    weights = [(w, g["{}/bias{}".format(name, i)]) for i, (w, _) in enumerate(weights)]
    return g, specs, weights


@pytest.mark.parametrize("name", ["toy", "real"])
def test_oracle_matches_committed_golden(name):
    g, specs, weights = rebuild_case(name)
    r = o.loss_and_gradients(specs, weights, g[name + "/x"].astype(np.float64), g[name + "/labels"],
                             g[name + "/prediction_lengths"], g[name + "/label_lengths"])
    np.testing.assert_allclose(r["probs"], g[name + "/probs"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(r["losses"], g[name + "/losses"], rtol=1e-9)
    for i, (dw, db) in enumerate(r["grads"]):
        np.testing.assert_allclose(np.linalg.norm(dw), g["{}/dw_norm{}".format(name, i)], rtol=1e-8)
        np.testing.assert_allclose(db, g["{}/db{}".format(name, i)], rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(dw[0, :4, :4], g["{}/dw_slice{}".format(name, i)], rtol=1e-7, atol=1e-12)
    dec = o.greedy_decode_indices(r["probs"], g[name + "/prediction_lengths"])
    for i, d in enumerate(dec):
        assert d == list(g[name + "/decoded"][i][:g[name + "/decoded_lengths"][i]])


def test_same_padding_rule():
    assert o.same_padding(1000, 48, 2) == (500, 23, 23)
    assert o.same_padding(999, 48, 2) == (500, 23, 24)
    assert o.same_padding(500, 32, 1) == (500, 15, 16)
    assert o.same_padding(500, 7, 1) == (500, 3, 3)
    assert o.same_padding(500, 1, 1) == (500, 0, 0)


def test_real_topology_counts():
    specs = o.layer_specs(128, 29)
    assert [s.name for s in specs] == ["striding_conv"] + ["inner_conv_%d" % i for i in range(1, 8)] + [
        "big_conv_1", "big_conv_2", "output_conv"]
    params = sum(s.kernel_size * s.cin * s.cout + s.cout for s in specs)
    assert params == 24662529
    flops = 2 * 500 * sum(s.kernel_size * s.cin * s.cout for s in specs)
    assert abs(flops / 1e9 - 24.6565) < 1e-3


def test_against_torch_autograd():
    specs = toy_specs()
    weights = o.glorot_uniform_weights(specs, 2, np.float64)
    weights = [(w, np.random.RandomState(5 + i).randn(*b.shape) * 0.1) for i, (w, b) in enumerate(weights)]
    x = np.random.RandomState(0).randn(3, 17, 4)
    labels = np.array([[0, 1, 2], [3, 3, -1], [-1, -1, -1]])
    pl, ll = [8, 7, 5], [3, 2, 0]
    r = o.loss_and_gradients(specs, weights, x, labels, pl, ll)
    t = tc.loss_and_gradients(specs, weights, x, labels, pl, ll)
    np.testing.assert_allclose(r["losses"], t["losses"], rtol=2e-6)
    for (dw, db), (tw, tb) in zip(r["grads"], t["grads"]):
        assert np.abs(dw - tw).max() <= 5e-6 * np.abs(dw).max()
        assert np.abs(db - tb).max() <= 5e-6 * np.abs(db).max()


def test_ctc_against_brute_force_enumeration():
    rng = np.random.RandomState(3)
    k, blank = 4, 3
    for t_len, label in [(4, [0, 1]), (5, [2, 2]), (5, [0, 1, 0]), (3, []), (6, [1, 1, 1]), (2, [0, 1, 2])]:
        probs = rng.dirichlet(np.ones(k), size=t_len)
        loss, _ = o.ctc_single(o.ctc_log_q(probs), label, blank)
        bf = o.ctc_brute_force(probs, label, blank)
        if np.isinf(bf):
            assert np.isinf(loss)
        else:
            assert abs(loss - bf) < 1e-10


def test_ctc_gradient_by_finite_differences():
    rng = np.random.RandomState(4)
    probs = rng.dirichlet(np.ones(5), size=(1, 7))
    labels = np.array([[1, 1, 2]])
    losses, dprobs = o.ctc_batch_cost(probs, labels, [6], [3])
    h = 1e-6
    for (t, c) in [(0, 1), (3, 4), (5, 2), (6, 0)]:
        p2 = probs.copy()
        p2[0, t, c] += h
        l2, _ = o.ctc_batch_cost(p2, labels, [6], [3])
        p3 = probs.copy()
        p3[0, t, c] -= h
        l3, _ = o.ctc_batch_cost(p3, labels, [6], [3])
        fd = (l2[0] - l3[0]) / (2 * h)
        assert abs(fd - dprobs[0, t, c]) < 1e-5 * max(1.0, abs(fd))
    assert np.all(dprobs[0, 6:] == 0)  # frames >= prediction_length are never scored


def test_weight_gradient_by_finite_differences():
    specs = toy_specs()
    weights = o.glorot_uniform_weights(specs, 2, np.float64)
    x = np.random.RandomState(0).randn(2, 12, 4)
    labels = np.array([[0, 1], [2, -1]])
    args = (labels, [6, 5], [2, 1])
    r = o.loss_and_gradients(specs, weights, x, *args)
    h = 1e-6
    for li, idx in [(0, (2, 1, 3)), (2, (1, 0, 5)), (3, (3, 2, 1)), (5, (0, 7, 4))]:
        def loss_at(delta):
            w2 = [(w.copy(), b.copy()) for w, b in weights]
            w2[li][0][idx] += delta
            return o.loss_and_gradients(specs, w2, x, *args)["mean_loss"]
        fd = (loss_at(h) - loss_at(-h)) / (2 * h)
        assert abs(fd - r["grads"][li][0][idx]) < 1e-6 + 1e-5 * abs(fd)


def test_frozen_layers_get_no_gradient():
    specs = toy_specs()
    weights = o.glorot_uniform_weights(specs, 2, np.float64)
    x = np.random.RandomState(0).randn(2, 12, 4)
    full = o.loss_and_gradients(specs, weights, x, np.array([[0, 1], [2, -1]]), [6, 5], [2, 1])
    r = o.loss_and_gradients(specs, weights, x, np.array([[0, 1], [2, -1]]), [6, 5], [2, 1], frozen_layer_count=3)
    for li in range(3):
        assert not r["grads"][li][0].any() and not r["grads"][li][1].any()
    for li in range(3, len(specs)):
        np.testing.assert_allclose(r["grads"][li][0], full["grads"][li][0])


def test_bf16_rounding_is_round_to_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.1415927, 1e-40, 65504.0], dtype=np.float32)
    r = o.round_to_bf16(x)
    import torch
    np.testing.assert_array_equal(r, torch.tensor(x).to(torch.bfloat16).to(torch.float32).numpy())


def test_ctc_core_against_tensorflow_known_answers():
    """tf.nn.ctc_loss / tf.nn.ctc_greedy_decoder known-answer vectors from TensorFlow's own unit tests (the third-party ops
    net.py:402-406 and net.py:452-454 bottom out in; provenance in the fixture): loss, gradient w.r.t. the logits and the
    decoded sequences, plus Keras' own backend test of K.ctc_batch_cost on the same data (padded batch, log(p + 1e-8))."""
    import json
    kat = json.loads((GOLDEN / "tf_known_answers.json").read_text())
    for case in kat["ctc_loss"]:
        p = np.array(case["probs"], dtype=np.float64)
        assert np.abs(p.sum(axis=1) - 1).max() < 2e-6  # the fixture's self-check: rows are softmax outputs
        k = p.shape[1]
        loss, du = o.ctc_single(np.log(p / p.sum(axis=1, keepdims=True)), case["labels"], k - 1)
        assert abs(loss - case["loss"]) < 1e-5
        assert np.abs(du - np.array(case["grad_logits"])).max() < 2e-6
        # the same through the Keras entry point (eps = 0 turns the re-softmax into the identity)
        losses, dp = o.ctc_batch_cost(p[None], np.array([case["labels"]]), [p.shape[0]], [len(case["labels"])], eps=0.0)
        assert abs(losses[0] - case["loss"]) < 1e-5
        assert np.abs(dp[0] * p - np.array(case["grad_logits"])).max() < 2e-6
        # ... and with the eps Keras adds the answer moves by ~1e-7 only
        losses, _ = o.ctc_batch_cost(p[None], np.array([case["labels"]]), [p.shape[0]], [len(case["labels"])])
        assert abs(losses[0] - case["loss"]) < 1e-5
    # Keras' own test of the wrapper (backend_test.py::test_ctc): both sequences as one padded batch
    kb = kat["keras_ctc_batch_cost"]
    probs = np.stack([np.array(c["probs"]) for c in kat["ctc_loss"]])
    losses, _ = o.ctc_batch_cost(probs, np.array(kb["labels"]), kb["input_lengths"], kb["label_lengths"])
    np.testing.assert_allclose(losses, kb["loss_log_probs_tf"], atol=kb["atol"])
    dec = kat["ctc_greedy_decoder"]
    got = o.greedy_decode_indices(np.array(dec["probs"]), dec["sequence_length"])
    assert [list(d) for d in got] == dec["decoded"]


def _tf_conv_cases():
    """(name, x (B,T,Cin), w (k,Cin,Cout), stride, expected (B,frames,Cout)) from the TensorFlow conv known answers"""
    import json
    cases = {c["name"]: c for c in json.loads((GOLDEN / "tf_known_answers.json").read_text())["conv"]["cases"]}
    out = []
    c = cases["testConv2D1x2Filter"]
    x = np.arange(1, 19, dtype=np.float64).reshape(c["x_shape"])
    w = np.arange(1, 19, dtype=np.float64).reshape(c["w_shape"])
    out.append((c["name"], x, w, 1, np.array(c["expected"]).reshape(2, 2, 3)))
    c = cases["testConv2D2x2FilterStride2Same"]
    x2 = np.arange(1, 19, dtype=np.float64).reshape(c["x_shape_2d"])
    w2 = np.arange(1, 37, dtype=np.float64).reshape(c["w_shape_2d"])
    x1 = np.concatenate([x2[0], x2[1]], axis=1)[None]   # (1, 3 frames, 6 = (kh, ci))
    w1 = np.concatenate([w2[0], w2[1]], axis=1)         # (kw, 6, co)
    out.append((c["name"], x1, w1, 2, np.array(c["expected"]).reshape(1, 2, 3)))
    return out, cases


def test_conv_against_tensorflow_known_answers():
    """TF "SAME" padding + convolution arithmetic of the oracle against TensorFlow's own conv_ops_test.py cases (fixture
    tests/golden/tf_known_answers.json): kernel 2 at stride 1, kernel 2 at stride 2 with the asymmetric right pad, and
    kernel 1 at stride 2."""
    convs, cases = _tf_conv_cases()
    for name, x, w, stride, want in convs:
        z = o.conv1d_preactivation(x, w, np.zeros(w.shape[2]), stride)
        assert np.array_equal(z[:, :want.shape[1]], want), name
    c = cases["testConv2DKernelSmallerThanStrideSame"]
    for n, want in zip(c["n"], c["expected"]):
        img = np.arange(1, n * n + 1, dtype=np.float64).reshape(n, n, 1)
        z = o.conv1d_preactivation(img, np.ones((1, 1, 1)), np.zeros(1), 2)
        assert z[::2].reshape(-1).tolist() == want


def test_conv_gradients_against_tensorflow_known_answers():
    """Input and filter gradients of the oracle's convolution against conv_ops_test.py's backprop known answers."""
    import json
    c = json.loads((GOLDEN / "tf_known_answers.json").read_text())["conv"]["backprop"]
    x, w, dz = np.array(c["x"])[None], np.array(c["w"]), np.array(c["dz"])[None]
    # the fixture's 1-D restatement is the published 2-D answer, transposed: dx[w][kh], dw[kw][kh]
    assert np.array(c["dx"]).T.reshape(-1).tolist() == c["expected_input_gradient_2d"]
    assert np.array(c["dw"])[:, :, 0].T.reshape(-1).tolist() == c["expected_filter_gradient_2d"]
    dx, dw, db = o.conv1d_backward(x, w, 1, dz)
    assert np.array_equal(dx[0], np.array(c["dx"])) and np.array_equal(dw, np.array(c["dw"])) and db.tolist() == [3.0]


def test_greedy_decode_known_answers():
    import json
    kat = json.loads((GOLDEN / "codec_golden.json").read_text(encoding="utf8"))["tf_greedy_kat"]
    logits = np.array(kat["logits_t_k"])[None]  # (1,T,K); argmax of logits == argmax of softmax
    assert o.greedy_decode_indices(logits, [5], blank=1) == [kat["greedy_merge_repeated"]]
    assert o.greedy_decode_indices(logits, [5], blank=1, merge_repeated=False) == [kat["greedy_no_merge"]]


def test_keras_adam_first_steps():
    p = np.array([1.0, -2.0])
    g = np.array([0.5, -0.25])
    m = np.zeros(2)
    v = np.zeros(2)
    p1, m1, v1 = o.keras_adam_step(p, g, m, v, 1)
    # at step 1 the bias-corrected update is lr * sign(g) (up to epsilon)
    np.testing.assert_allclose(p - p1, 1e-4 * np.sign(g), rtol=1e-5)


def test_bf16_storage_noise_is_in_the_activations_not_in_g():
    """Where the bf16 path's gradient error comes from (HISTORY.md section 1 "bf16 gradient noise"), on the float64 oracle run with
    the HIP path's rounding points: keeping the back-propagated signal g in fp32 instead of bf16 changes the per-layer
    weight-gradient error by less than a tenth of it -- the error is made in the forward pass (bf16 activations flip the
    sign of pre-activations within rounding of zero, one whole element of g per flip), it grows from 1e-3 at the output
    layer to > 5 % at the bottom of the stack, and the loss is untouched (1e-5)."""
    k, f, b, t = 29, 128, 2, 64
    specs = o.layer_specs(f, k)
    rng = np.random.RandomState(100)
    weights = [(w.astype(np.float64), rng.uniform(-.05, .05, size=bb.shape))
               for w, bb in o.glorot_uniform_weights(specs, seed=2, dtype=np.float32)]
    x = np.random.RandomState(0).randn(b, t, f)
    lr = np.random.RandomState(1)
    lab_len = [9, 7]
    labels = o.pack_label_batch([list(lr.randint(0, k - 1, size=n)) for n in lab_len])
    pred = [32, 31]
    ref = o.loss_and_gradients(specs, weights, x, labels, pred, lab_len)
    err = {}
    for mode in (True, "fp32_g"):
        r = o.loss_and_gradients(specs, weights, x, labels, pred, lab_len, bf16_mirror=mode)
        assert np.abs(r["losses"] / ref["losses"] - 1).max() < 1e-4
        err[mode] = np.array([np.linalg.norm(g[0] - rg[0]) / np.linalg.norm(rg[0])
                              for g, rg in zip(r["grads"], ref["grads"])])
    assert np.all(np.abs(err[True] - err["fp32_g"]) < 0.1 * err[True] + 1e-4), (err[True], err["fp32_g"])
    assert err[True][-1] < 5e-3 and err[True][0] > 5e-2
    assert np.all(err[True][:-1] > err[True][1:] * 0.8)  # grows from the output layer downwards


def test_float64_torch_realisation_against_the_numpy_oracle():
    """oracle/w2l_float64.py (per-tap matmul + F.ctc_loss in float64: the exact corner of bench.py's parity triangle)
    against the numpy restatement on a shrunken stack with ragged lengths and repeated labels: losses, every gradient and
    the ReLU decisions agree to float64 rounding."""
    from oracle import w2l_float64 as f64
    specs = o.layer_specs(12, 6, main_filter_count=10, out_filter_count=16, inner_count=2, striding_kernel=6,
                          inner_kernel=3, big_kernel=4)
    weights = o.glorot_uniform_weights(specs, 1, dtype=np.float64)
    rng = np.random.RandomState(0)
    weights = [(w, rng.uniform(-0.1, 0.1, size=b.shape)) for w, b in weights]
    x = rng.randn(3, 21, 12)
    labels = o.pack_label_batch([[0, 1, 1], [2], [3, 4, 0, 0]])
    pred_len, lab_len = [11, 9, 11], [3, 1, 4]
    a = o.loss_and_gradients(specs, weights, x, labels, pred_len, lab_len)
    b = f64.loss_and_gradients(specs, weights, x, labels, pred_len, lab_len)
    np.testing.assert_allclose(b["losses"], a["losses"], rtol=1e-12)
    np.testing.assert_allclose(b["probs"], a["probs"], atol=1e-14)
    for (aw, ab), (bw, bb) in zip(a["grads"], b["grads"]):
        np.testing.assert_allclose(bw, aw, atol=1e-13 * np.abs(aw).max())
        np.testing.assert_allclose(bb, ab, atol=1e-13 * max(np.abs(ab).max(), 1e-300))
    _, _, zs = o.forward_stack(specs, weights, x, keep=True)
    for z, m in zip(zs[:-1], b["masks"]):
        assert np.array_equal(z > 0, m.numpy())
