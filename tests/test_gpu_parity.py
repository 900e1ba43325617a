"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the CPU oracle on
the same seeded inputs; bit-exact for integer outputs (decoded indices), stated tolerances for floating point.

Tolerances (relative to the oracle, fp32 reference semantics of Keras/TF):
  f32 path : activations 1e-4 of layer max; loss 1e-5; gradients 1e-4 rel-L2                (north_star: 1e-3)
  bf16 path: vs the oracle run with the SAME storage rounding points (bf16_mirror): loss 1e-3, gradients 2e-2 rel-L2;
             vs the pure fp32 oracle: loss 1e-3 (north_star config 3), gradients reported in gpurun_out/parity.json.
"""
import ctypes
import json
import os
from pathlib import Path

import numpy as np
import pytest

from oracle import w2l_oracle as o

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
REPORT = {}


def _report(key, value):
    REPORT[key] = value
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "parity.json").write_text(json.dumps(REPORT, indent=1, sort_keys=True, default=float))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def make_case(b=2, t=64, f=128, k=29, seed=0, sizes=None, bias_scale=0.05):
    from speechless_amd.engine import wav2letter_layer_specs
    sizes = sizes or {}
    specs = wav2letter_layer_specs(f, k, **sizes)
    ospecs = o.layer_specs(f, k, **sizes)
    weights = o.glorot_uniform_weights(ospecs, seed=2, dtype=np.float32)
    rng = np.random.RandomState(100 + seed)
    weights = [(w, rng.uniform(-bias_scale, bias_scale, size=bb.shape).astype(np.float32)) for (w, bb) in weights]
    x = np.random.RandomState(seed).randn(b, t, f).astype(np.float32)
    t_out = -(-t // 2)
    lrng = np.random.RandomState(seed + 1)
    lab_len = [int(lrng.randint(1, max(2, min(t_out // 3, 200)))) for _ in range(b)]
    labels = [list(lrng.randint(0, k - 1, size=n)) for n in lab_len]
    pred_len = [t // 2 - (i % 3) for i in range(b)]
    return dict(specs=specs, ospecs=ospecs, weights=weights, x=x, labels=o.pack_label_batch(labels),
                label_lengths=lab_len, prediction_lengths=pred_len, k=k, t_out=t_out)


def make_engine(case, dtype, **kw):
    from speechless_amd.engine import Engine
    eng = Engine(case["specs"], case["k"], dtype=dtype, **kw)
    eng.set_weights(case["weights"])
    return eng


def layer_activation(eng, buf, index):
    from speechless_amd.engine import HALO
    s = eng.specs[index]
    y = buf.y[index].float().cpu().numpy()
    return y[:, HALO:HALO + buf.t_out, :s.cout], y


def weights64(case):
    return [(w.astype(np.float64), b.astype(np.float64)) for w, b in case["weights"]]


# ------------------------------------------------------------------------------------------ forward
@pytest.mark.parametrize("t", [64, 77])
def test_forward_f32_layer_by_layer(t):
    import torch
    from speechless_amd.engine import HALO
    case = make_case(b=3, t=t)
    eng = make_engine(case, "f32")
    probs = eng.forward(case["x"]).cpu().numpy()
    torch.cuda.synchronize()
    buf = eng.cur
    ref_probs, xs, zs = o.forward_stack(case["ospecs"], weights64(case), case["x"].astype(np.float64), keep=True)
    for i in range(len(case["specs"]) - 1):
        got, raw = layer_activation(eng, buf, i)
        want = np.maximum(zs[i], 0)
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err < 1e-4, "layer {} ({}) activation error {}".format(i, case["specs"][i].name, err)
        # layout invariants: halo rows, rows past the valid time and padded channels stay zero -- except the last padded
        # channel of a hidden layer, which carries the constant 1 on every valid frame (Engine.ones_channel)
        assert not raw[:, :HALO].any() and not raw[:, HALO + buf.t_out:].any()
        padding = raw[:, :, case["specs"][i].cout:]
        if eng.ones_channel and padding.shape[2]:
            assert (padding[:, HALO:HALO + buf.t_out, -1] == 1).all()
            padding = padding[:, :, :-1]
        assert not padding.any()
    assert np.abs(probs - ref_probs).max() < 1e-5
    np.testing.assert_allclose(probs.sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_conv_kernels_against_tensorflow_known_answers(dtype):
    """The HIP convolution (both dtypes; fp32 accumulate) on TensorFlow's own conv_ops_test.py known answers
    (tests/golden/tf_known_answers.json): kernel 2 / stride 1 and kernel 2 / stride 2 with SAME's asymmetric right pad.
    Integer data: the fp32 path must be exact, the bf16 path exact up to the bf16 rounding of the stored activation."""
    from speechless_amd.engine import Engine, LayerSpec
    from test_oracle import _tf_conv_cases
    convs, _ = _tf_conv_cases()
    for name, x, w, stride, want in convs:
        cin, cout = w.shape[1], w.shape[2]
        specs = [LayerSpec("striding_conv", w.shape[0], stride, cin, cout, "relu"),
                 LayerSpec("output_conv", 1, 1, cout, 4, "softmax")]
        eng = Engine(specs, 4, dtype=dtype)
        eng.set_weights([(w.astype(np.float32), np.zeros(cout, dtype=np.float32)),
                         (np.zeros((1, cout, 4), dtype=np.float32), np.zeros(4, dtype=np.float32))])
        eng.forward(x.astype(np.float32))
        got, _ = layer_activation(eng, eng.cur, 0)   # relu(z); every expected value is positive
        got = got[:, :want.shape[1]]
        if dtype == "f32":
            assert np.array_equal(got, want), (name, got)
        else:
            assert np.array_equal(got, o.round_to_bf16(want.astype(np.float32))), (name, got)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_conv_gradient_kernels_against_tensorflow_known_answers(dtype):
    """sl_conv1d_wgrad / sl_bias_grad / the dgrad launch of sl_conv1d_nt on conv_ops_test.py's backprop known answers
    (tests/golden/tf_known_answers.json -> conv.backprop): small integers, exact in both dtypes."""
    import ctypes
    import torch
    from speechless_amd import _lib
    from speechless_amd.engine import Engine, LayerSpec, HALO
    c = json.loads((ROOT / "tests" / "golden" / "tf_known_answers.json").read_text())["conv"]["backprop"]
    x, w, dz = np.array(c["x"], dtype=np.float32), np.array(c["w"], dtype=np.float32), np.array(c["dz"], dtype=np.float32)
    specs = [LayerSpec("striding_conv", 2, 2, 4, 2, "relu"), LayerSpec("inner_conv_1", 2, 1, 2, 1, "relu"),
             LayerSpec("output_conv", 1, 1, 1, 4, "softmax")]
    eng = Engine(specs, 4, dtype=dtype)
    eng.set_weights([(np.zeros((2, 4, 2), dtype=np.float32), np.zeros(2, dtype=np.float32)), (w, np.zeros(1, dtype=np.float32)),
                     (np.zeros((1, 1, 4), dtype=np.float32), np.zeros(4, dtype=np.float32))])
    buf = eng.load_input(np.zeros((1, 6, 4), dtype=np.float32))   # 6 input frames -> 3 frames after the stride
    buf.ensure_backward(eng)
    eng.repack_weights()
    assert buf.t_out == 3
    td = eng.torch_dtype
    y0 = torch.zeros_like(buf.y[0])
    y0[0, HALO:HALO + 3, :2] = torch.tensor(x).to(td)
    buf.y[0].copy_(y0)
    g1 = torch.zeros_like(buf.g[1])
    g1[0, HALO:HALO + 3, :1] = torch.tensor(dz).to(td)
    buf.g[1].copy_(g1)
    st = torch.cuda.current_stream().cuda_stream
    p = eng.plans[1]
    dw_v, db_v = eng.layer_param_views(eng.grads, p)
    eng.lib.call("sl_conv1d_wgrad", buf.y[0].data_ptr(), buf.g[1].data_ptr(), dw_v.data_ptr(),
                 ctypes.byref(buf.wgrad_geom[1]), eng.dtype_code, 0, buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)
    eng.lib.call("sl_bias_grad", buf.g[1].data_ptr(), db_v.data_ptr(), ctypes.byref(buf.wgrad_geom[1]), eng.dtype_code,
                 buf.bias_ws.data_ptr(), buf.bias_ws.numel(), st)
    eng.lib.call("sl_conv1d_nt", buf.g[1].data_ptr(), eng.w_dgrad[1].data_ptr(), None, buf.y[0].data_ptr(),
                 buf.g[0].data_ptr(), ctypes.byref(buf.dgrad_geom[1]), _lib.EPI_RELU_MASK, eng.dtype_code, 0, 0,
                 buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
    torch.cuda.synchronize()
    assert np.array_equal(dw_v[:, :2, :1].cpu().numpy(), np.array(c["dw"]))
    assert db_v[:1].cpu().numpy().tolist() == [3.0]
    assert np.array_equal(buf.g[0][0, HALO:HALO + 3, :2].float().cpu().numpy(), np.array(c["dx"]))


def test_forward_bf16_matches_mirrored_oracle():
    import torch
    case = make_case(b=3, t=64)
    eng = make_engine(case, "bf16")
    probs = eng.forward(case["x"]).cpu().numpy()
    torch.cuda.synchronize()
    w32 = case["weights"]
    ref_probs, xs, zs = o.forward_stack(case["ospecs"], w32, case["x"], bf16_mirror=True, keep=True)
    worst = 0.0
    for i in range(len(case["specs"]) - 1):
        got, _ = layer_activation(eng, eng.cur, i)
        want = o.round_to_bf16(np.maximum(zs[i], 0))
        err = np.abs(got - want).max() / np.abs(want).max()
        worst = max(worst, err)
        assert err < 2e-2, "layer {} bf16 activation error {}".format(i, err)  # 1-2 bf16 ulps of the layer max
    _report("fwd_bf16_vs_mirror_worst_layer_err", worst)
    assert np.abs(probs - ref_probs).max() < 2e-3
    exact = o.forward_stack(case["ospecs"], weights64(case), case["x"].astype(np.float64))
    _report("fwd_bf16_vs_fp32_oracle_max_prob_err", float(np.abs(probs - exact).max()))


def test_greedy_decode_bit_exact_f32_full_length():
    """BASELINE config 2 semantics at 1000 frames (smaller batch so the float64 oracle finishes in seconds):
    frame argmax and decoded label indices must equal the CPU oracle's exactly."""
    case = make_case(b=2, t=1000, seed=3)
    eng = make_engine(case, "f32")
    probs = eng.forward(case["x"]).cpu().numpy()
    pred_len = [500, 497]
    decoded, frame_argmax = eng.greedy_decode(pred_len)
    ref = o.forward_stack(case["ospecs"], weights64(case), case["x"].astype(np.float64))
    ref_idx, margin = o.frame_argmax_and_margin(ref)
    _report("config2_min_top1_top2_margin", float(margin.min()))
    for i, n in enumerate(pred_len):
        mism = np.nonzero(frame_argmax[i, :n] != ref_idx[i, :n])[0]
        assert mism.size == 0, "argmax differs at frames {} (margins {})".format(mism[:5], margin[i, mism[:5]])
        assert (frame_argmax[i, n:] == -1).all()
    assert decoded == o.greedy_decode_indices(ref, pred_len)
    # and the GPU decoder on the GPU's own probabilities agrees with the numpy twin on them
    assert decoded == o.greedy_decode_indices(probs, pred_len)


# ------------------------------------------------------------------------------------------ loss + gradients
def run_loss_and_grads(eng, case):
    import torch
    eng.load_input(case["x"])
    eng.set_labels(case["labels"], case["label_lengths"], case["prediction_lengths"])
    eng.forward()
    loss = eng.ctc()
    eng.backward()
    torch.cuda.synchronize()
    return loss.cpu().numpy(), eng.get_gradients()


@pytest.mark.parametrize("t", [64, 77])
def test_loss_and_gradients_f32(t):
    case = make_case(b=3, t=t, seed=5)
    eng = make_engine(case, "f32")
    losses, grads = run_loss_and_grads(eng, case)
    ref = o.loss_and_gradients(case["ospecs"], weights64(case), case["x"].astype(np.float64), case["labels"],
                               case["prediction_lengths"], case["label_lengths"])
    np.testing.assert_allclose(losses, ref["losses"], rtol=1e-5)
    worst = 0.0
    for i, ((dw, db), (rw, rb)) in enumerate(zip(grads, ref["grads"])):
        ew, eb = rel_l2(dw, rw), rel_l2(db, rb)
        worst = max(worst, ew, eb)
        assert ew < 1e-4 and eb < 1e-4, "layer {}: dW rel-L2 {}, db rel-L2 {}".format(i, ew, eb)
    _report("grads_f32_worst_rel_l2_t{}".format(t), worst)


def test_loss_and_gradients_bf16():
    case = make_case(b=3, t=64, seed=5)
    eng = make_engine(case, "bf16")
    losses, grads = run_loss_and_grads(eng, case)
    args = (case["labels"], case["prediction_lengths"], case["label_lengths"])
    mirror = o.loss_and_gradients(case["ospecs"], case["weights"], case["x"], *args, bf16_mirror=True)
    exact = o.loss_and_gradients(case["ospecs"], weights64(case), case["x"].astype(np.float64), *args)
    np.testing.assert_allclose(losses, mirror["losses"], rtol=1e-3)
    np.testing.assert_allclose(losses, exact["losses"], rtol=1e-3)  # north_star config 3: loss within 1e-3 of CPU
    rep = {}
    for i, ((dw, db), (mw, mb), (xw, xb)) in enumerate(zip(grads, mirror["grads"], exact["grads"])):
        rep[case["specs"][i].name] = dict(dw_vs_mirror=rel_l2(dw, mw), db_vs_mirror=rel_l2(db, mb),
                                          dw_vs_fp32=rel_l2(dw, xw), db_vs_fp32=rel_l2(db, xb))
    _report("grads_bf16", rep)
    # bf16 storage of the back-propagated signal is inherently noisy in a random-init net: the CPU oracle's own
    # bf16 mirror deviates from exact fp32 by 0.1 % (output_conv) ... 16 % (striding_conv), and a 1e-6 perturbation of
    # the weights moves the mirrored layer-0 gradient by 12 % (rounding decisions flip; measured in DESIGN.md).  The
    # kernels themselves are exact (test_single_layer_kernels_with_exact_operands).  Criterion: the HIP path must be
    # no noisier than the storage scheme itself.
    for i, name in enumerate(s.name for s in case["specs"]):
        r = rep[name]
        scheme_w = rel_l2(mirror["grads"][i][0], exact["grads"][i][0])
        scheme_b = rel_l2(mirror["grads"][i][1], exact["grads"][i][1])
        assert r["dw_vs_fp32"] < 1.5 * scheme_w + 2e-3, (name, r, scheme_w)
        assert r["db_vs_fp32"] < 1.5 * scheme_b + 2e-3, (name, r, scheme_b)
    for name, tol in (("output_conv", 2e-3), ("big_conv_2", 1e-2), ("big_conv_1", 1.5e-2)):  # 81 % of the parameters
        assert rep[name]["dw_vs_mirror"] < tol, (name, rep[name])
    _report("loss_bf16_rel_err_vs_fp32", float(np.abs(losses / exact["losses"] - 1).max()))


def test_frozen_layers_are_skipped():
    case = make_case(b=2, t=64, seed=6)
    eng = make_engine(case, "f32", frozen_layer_count=8)
    _, grads = run_loss_and_grads(eng, case)
    ref = o.loss_and_gradients(case["ospecs"], weights64(case), case["x"].astype(np.float64), case["labels"],
                               case["prediction_lengths"], case["label_lengths"], frozen_layer_count=8)
    for i, ((dw, db), (rw, rb)) in enumerate(zip(grads, ref["grads"])):
        if i < 8:
            assert not dw.any() and not db.any()
        else:
            assert rel_l2(dw, rw) < 1e-4


def test_backward_is_deterministic():
    import torch
    case = make_case(b=4, t=128, seed=7)
    eng = make_engine(case, "bf16")
    run_loss_and_grads(eng, case)
    first = eng.grads.clone()
    run_loss_and_grads(eng, case)
    assert torch.equal(first, eng.grads)


# ------------------------------------------------------------------------------------------ CTC kernel alone
def run_ctc_kernel(hip_lib, logits, labels, label_len, input_len, eps=1e-8):
    import torch
    from speechless_amd import _lib
    b, t, k = logits.shape
    dev = "cuda:0"
    lg = torch.tensor(logits, dtype=torch.float32, device=dev)
    probs = torch.zeros((b, t, k), dtype=torch.float32, device=dev)
    logq = torch.zeros_like(probs)
    lab = torch.tensor(labels, dtype=torch.int32, device=dev)
    ll = torch.tensor(label_len, dtype=torch.int32, device=dev)
    il = torch.tensor(input_len, dtype=torch.int32, device=dev)
    loss = torch.zeros((b,), dtype=torch.float32, device=dev)
    dl = torch.zeros((b, t, k), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    hip_lib.call("sl_softmax_logq", lg.data_ptr(), probs.data_ptr(), logq.data_ptr(), b, t, k, k, t * k, eps, st)
    need = hip_lib.raw("sl_ctc_workspace_bytes")(b, t, lab.shape[1])
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    hip_lib.call("sl_ctc_loss_grad", probs.data_ptr(), logq.data_ptr(), lab.data_ptr(), ll.data_ptr(), il.data_ptr(),
                 loss.data_ptr(), dl.data_ptr(), b, t, k, lab.shape[1], 0, k, t * k, _lib.SL_F32, eps, 1.0,
                 ws.data_ptr(), need, st)
    torch.cuda.synchronize()
    return probs.cpu().numpy(), loss.cpu().numpy(), dl.cpu().numpy()


def test_ctc_kernel_edge_cases(hip_lib):
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
    np.testing.assert_allclose(probs, ref_p, atol=2e-7)
    ref_loss, ref_dp = o.ctc_batch_cost(ref_p, labels, input_len, lab_len)
    ref_dl = o.softmax_backward(ref_p, ref_dp)
    assert np.isinf(ref_loss[5]) and np.isinf(loss[5])
    np.testing.assert_allclose(loss[:5], ref_loss[:5], rtol=1e-5)
    for i in range(len(labels_list)):
        # fp32 log2-domain lattice with raw v_exp_f32/v_log_f32 (1 ulp) vs the float64 oracle; gradient entries are O(1)
        assert np.abs(dl[i] - ref_dl[i]).max() < 1e-4, (i, np.abs(dl[i] - ref_dl[i]).max())
        assert not dl[i, input_len[i]:].any()


def test_ctc_kernel_long_labels(hip_lib):
    rng = np.random.RandomState(10)
    k, t, b = 29, 500, 4
    lab_len = [200, 137, 1, 60]
    labels_list = [list(rng.randint(0, 28, size=n)) for n in lab_len]
    logits = rng.randn(b, t, k).astype(np.float32)
    labels = o.pack_label_batch(labels_list)
    input_len = [500, 480, 500, 333]
    _, loss, dl = run_ctc_kernel(hip_lib, logits, labels, lab_len, input_len)
    ref_p = o.softmax(logits.astype(np.float64))
    ref_loss, ref_dp = o.ctc_batch_cost(ref_p, labels, input_len, lab_len)
    np.testing.assert_allclose(loss, ref_loss, rtol=1e-5)
    assert rel_l2(dl, o.softmax_backward(ref_p, ref_dp)) < 1e-3  # fp32 log-space lattice (|alpha| ~ 1e3, ulp 1e-4) vs float64 oracle


def test_ctc_and_decode_kernels_against_tensorflow_known_answers(hip_lib):
    """The HIP softmax + CTC lattice / gradient kernels and the greedy decoder on TensorFlow's own known-answer vectors
    for tf.nn.ctc_loss / tf.nn.ctc_greedy_decoder (tests/golden/tf_known_answers.json): not via the oracle."""
    import torch
    kat = json.loads((ROOT / "tests" / "golden" / "tf_known_answers.json").read_text())
    cases = kat["ctc_loss"]
    t, k = 5, 6
    logits = np.stack([np.log(np.array(c["probs"], dtype=np.float64)) for c in cases]).astype(np.float32)
    labels = -np.ones((len(cases), 5), dtype=np.int32)
    for i, c in enumerate(cases):
        labels[i, :len(c["labels"])] = c["labels"]
    lab_len = [len(c["labels"]) for c in cases]
    # eps = 0: the op's own arithmetic; 1e-8 with the padded batch: Keras' backend_test.py::test_ctc (atol 1e-5)
    assert labels.tolist() == kat["keras_ctc_batch_cost"]["labels"]
    for eps, tol in ((0.0, 3e-6), (1e-8, 5e-6)):
        probs, loss, dl = run_ctc_kernel(hip_lib, logits, labels, lab_len, [t, t], eps=eps)
        for i, c in enumerate(cases):
            assert np.abs(probs[i] - np.array(c["probs"])).max() < 2e-6
            assert abs(loss[i] - c["loss"]) < 1e-5, (eps, i, loss[i])
            # d loss / d (network logits) == TF's gradient w.r.t. its logits (the softmax chain contributes
            # -p * sum_k(du_k) = 0 because softmax - occupancy sums to zero per frame)
            assert np.abs(dl[i] - np.array(c["grad_logits"])).max() < tol, (eps, i)
    dec = kat["ctc_greedy_decoder"]
    p = torch.tensor(np.array(dec["probs"]), dtype=torch.float32, device="cuda:0")
    il = torch.tensor(dec["sequence_length"], dtype=torch.int32, device="cuda:0")
    out = torch.zeros((2, 6), dtype=torch.int32, device="cuda:0")
    out_len = torch.zeros((2,), dtype=torch.int32, device="cuda:0")
    hip_lib.call("sl_greedy_decode", p.data_ptr(), il.data_ptr(), out.data_ptr(), out_len.data_ptr(), None, 2, 6, 4, 3,
                 torch.cuda.current_stream().cuda_stream)
    got = [out[i, :int(out_len[i])].tolist() for i in range(2)]
    assert got == dec["decoded"]
    assert out[0].tolist() == [0, 1, -1, -1, -1, -1]


@pytest.mark.parametrize("k,t", [(29, 1000), (33, 130), (5, 77)])
def test_fused_output_softmax_matches_the_two_launch_path(k, t):
    """sl_output_softmax (1x1 output layer + softmax + log(p + eps) re-normalisation in one launch) against
    sl_conv1d_nt + sl_softmax_logq on the same activations; k = 33 exceeds the fused kernel's 32 classes and must fall
    back without a difference."""
    import torch
    case = make_case(b=3, t=t, k=k, seed=50 + k)
    outs = []
    for fuse in (False, True):
        eng = make_engine(case, "bf16")
        eng.fuse_output_softmax = fuse
        probs = eng.forward(case["x"]).cpu().numpy().copy()
        logq = eng.cur.logq.cpu().numpy().copy()
        torch.cuda.synchronize()
        outs.append((probs, logq))
    np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(outs[1][0].sum(-1), 1.0, atol=1e-5)
    if k > 32:
        assert np.array_equal(outs[0][0], outs[1][0])


def test_profile_hook_brackets_the_main_kernel():
    """sl_profile_next_kernel: the library records the two events around the main kernel of the NEXT conv call only."""
    import torch
    case = make_case(b=2, t=128, seed=5)
    eng = make_engine(case, "bf16")
    eng.forward(case["x"])
    torch.cuda.synchronize()
    eng.kernel_timeline = ({"fwd:big_conv_1"}, [])
    eng.forward(case["x"])
    torch.cuda.synchronize()
    recorded = eng.kernel_timeline[1]
    eng.kernel_timeline = None
    assert [t for t, _, _ in recorded] == ["fwd:big_conv_1"]
    ms = recorded[0][1].elapsed_time(recorded[0][2])
    assert 0.0 < ms < 5.0
    probs = eng.forward(case["x"]).cpu().numpy()  # the hook disarmed itself: a plain pass still works
    assert np.isfinite(probs).all()


# ------------------------------------------------------------------------------------------ decode, Adam
def test_greedy_decode_known_answers(hip_lib):
    import torch
    kat = json.loads((ROOT / "tests" / "golden" / "codec_golden.json").read_text(encoding="utf8"))["tf_greedy_kat"]
    probs = torch.tensor(o.softmax(np.array(kat["logits_t_k"])[None]), dtype=torch.float32, device="cuda:0")
    il = torch.tensor([5], dtype=torch.int32, device="cuda:0")
    out = torch.zeros((1, 5), dtype=torch.int32, device="cuda:0")
    out_len = torch.zeros((1,), dtype=torch.int32, device="cuda:0")
    hip_lib.call("sl_greedy_decode", probs.data_ptr(), il.data_ptr(), out.data_ptr(), out_len.data_ptr(), None, 1, 5,
                 2, 1, torch.cuda.current_stream().cuda_stream)
    assert int(out_len[0]) == 2 and out[0].tolist() == [0, 0, -1, -1, -1]  # test_ctc_decoders.py:40


def test_greedy_decode_random_and_ties(hip_lib):
    import torch
    rng = np.random.RandomState(11)
    b, t, k = 5, 333, 29
    probs = rng.dirichlet(np.ones(k) * 0.3, size=(b, t)).astype(np.float32)
    probs[0, :50] = 1.0 / k           # exact ties -> first index wins (numpy argmax semantics)
    probs[1, 10:200, :] = 0
    probs[1, 10:200, 28] = 1          # long blank run
    probs[2, :, :] = 0
    probs[2, :, 3] = 1                # one long repeat -> single symbol
    lens = [333, 300, 333, 0, 1]
    dev = "cuda:0"
    p = torch.tensor(probs, device=dev)
    il = torch.tensor(lens, dtype=torch.int32, device=dev)
    out = torch.zeros((b, t), dtype=torch.int32, device=dev)
    out_len = torch.zeros((b,), dtype=torch.int32, device=dev)
    fa = torch.zeros((b, t), dtype=torch.int32, device=dev)
    hip_lib.call("sl_greedy_decode", p.data_ptr(), il.data_ptr(), out.data_ptr(), out_len.data_ptr(), fa.data_ptr(), b,
                 t, k, k - 1, torch.cuda.current_stream().cuda_stream)
    ref = o.greedy_decode_indices(probs, lens)
    got = [out[i, :int(out_len[i])].tolist() for i in range(b)]
    assert got == ref
    assert (out[3] == -1).all()


def test_adam_matches_keras_formula(hip_lib):
    import torch
    rng = np.random.RandomState(12)
    n = 4096
    p = rng.randn(n).astype(np.float32)
    m = np.zeros(n, dtype=np.float32)
    v = np.zeros(n, dtype=np.float32)
    tp, tm, tv = (torch.tensor(a, device="cuda:0") for a in (p, m, v))
    rp, rm, rv = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for step in range(1, 4):
        g = rng.randn(n).astype(np.float32)
        tg = torch.tensor(g, device="cuda:0")
        hip_lib.call("sl_adam_step", tp.data_ptr(), tg.data_ptr(), tm.data_ptr(), tv.data_ptr(), n, step, 1e-4, 0.9,
                     0.999, 1e-8, torch.cuda.current_stream().cuda_stream)
        rp, rm, rv = o.keras_adam_step(rp, g.astype(np.float64), rm, rv, step)
    np.testing.assert_allclose(tp.cpu().numpy(), rp, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(tm.cpu().numpy(), rm, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(tv.cpu().numpy(), rv, rtol=2e-5, atol=1e-12)


# ------------------------------------------------------------------------------------------ full-size properties
def test_full_size_bf16_vs_f32_paths_agree():
    """BASELINE config 3 shape (B=32, 128 mel x 1000 frames, labels <= 200): the bf16 MFMA path and the independently
    written fp32 path (itself pinned to the oracle above) agree: loss within 1e-3, gradients within 2e-2 rel-L2,
    batch-order equivariance, zero-gradient rows for never-scored frames."""
    import torch
    case = make_case(b=32, t=1000, seed=21)
    results = {}
    for dtype in ("f32", "bf16"):
        eng = make_engine(case, dtype)
        losses, _ = run_loss_and_grads(eng, case)
        results[dtype] = (losses, eng.grads.clone(), eng.cur.probs.clone())
        del eng
        torch.cuda.empty_cache()
    lf, gf, pf = results["f32"]
    lb, gb, pb = results["bf16"]
    assert np.isfinite(lf).all()
    np.testing.assert_allclose(lb, lf, rtol=1e-3)
    err = float(torch.linalg.norm(gb - gf) / torch.linalg.norm(gf))
    _report("full_size_grad_rel_l2_bf16_vs_f32", err)
    _report("full_size_loss_rel_err_bf16_vs_f32", float(np.abs(lb / lf - 1).max()))
    agree = float((pb.argmax(-1) == pf.argmax(-1)).float().mean())
    _report("full_size_argmax_agreement_bf16_vs_f32", agree)
    assert err < 2e-2
    assert agree > 0.97


def test_full_size_batch_permutation_equivariance():
    import torch
    case = make_case(b=8, t=1000, seed=22)
    eng = make_engine(case, "bf16")
    l1, _ = run_loss_and_grads(eng, case)
    g1 = eng.grads.clone()
    perm = np.random.RandomState(0).permutation(8)
    case2 = dict(case)
    case2["x"] = case["x"][perm]
    case2["labels"] = case["labels"][perm]
    case2["label_lengths"] = [case["label_lengths"][i] for i in perm]
    case2["prediction_lengths"] = [case["prediction_lengths"][i] for i in perm]
    l2, _ = run_loss_and_grads(eng, case2)
    np.testing.assert_array_equal(l1[perm], l2)  # per-utterance work is independent -> bitwise
    assert float(torch.linalg.norm(eng.grads - g1) / torch.linalg.norm(g1)) < 1e-5  # sum order over b changes


# ------------------------------------------------------------------------------------------ API surface
def synthetic_examples(n, rng, f=128):
    from speechless_amd.net import LabeledSpectrogram
    words = ["she", "wasn't", "three", "abc", "xyz", "a", "it's", "zoo"]
    out = []
    for i in range(n):
        t = int(rng.randint(120, 200))
        label = " ".join(rng.choice(words, size=rng.randint(1, 4)))
        out.append(LabeledSpectrogram(id="utt{}".format(i), label=label, spectrogram=rng.randn(t, f)))
    return out


def test_wav2letter_api_train_predict_save_load(tmp_path):
    from speechless_amd import Wav2Letter, english_frequent_characters
    from speechless_amd.net import Adam
    rng = np.random.RandomState(13)
    net = Wav2Letter(128, english_frequent_characters, optimizer=Adam(1e-3), seed=1)
    assert net.input_to_prediction_length_ratio == 2
    batch = synthetic_examples(4, rng)
    before = net.test_and_predict_batch(batch)
    assert len(before.results) == 4 and all(isinstance(r.predicted, str) for r in before.results)
    assert isinstance(net.predict(batch[0]), str)
    assert np.isfinite(net.test_and_predict(batch[0]).loss)
    net.train([batch] * 12, preview_labeled_spectrogram_batch=batch[:2], tensor_board_log_directory=tmp_path / "tb",
              net_directory=tmp_path / "nets", batches_per_epoch=4)
    after = net.test_and_predict_batch(batch)
    assert after.average_loss < before.average_loss
    saved = sorted(p.name for p in (tmp_path / "nets").iterdir())
    assert saved and all(name.startswith("weights-epoch") for name in saved)
    net2 = Wav2Letter(128, english_frequent_characters, load_model_from_directory=tmp_path / "nets", load_epoch=2)
    again = net2.test_and_predict_batch(batch)
    assert [r.predicted for r in again.results] == [r.predicted for r in after.results]
    probs = net2.prediction_batch(np.stack([e.z_normalized_transposed_spectrogram()[:120] for e in batch]))
    assert probs.shape == (4, 60, 29)
    assert net2.predict_batch_greedily([e.z_normalized_transposed_spectrogram() for e in batch]) == \
        [r.predicted for r in again.results]


def test_constructor_errors_match_reference():
    from speechless_amd import Wav2Letter, english_frequent_characters
    with pytest.raises(ValueError):
        Wav2Letter(128, english_frequent_characters, frozen_layer_count=2)  # net.py:144-145
    with pytest.raises(NotImplementedError):
        Wav2Letter(128, english_frequent_characters, use_asg=True)  # net.py:396-399


# ------------------------------------------------------------------------------------------ single kernels, exact inputs
def _bf16_exact(rng, shape, scale=1.0):
    """Random values that are exactly representable in bf16 (so bf16 and fp32 kernels see identical operands)."""
    return o.round_to_bf16((rng.randn(*shape) * scale).astype(np.float32))


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("layer", [0, 1, 8, 9, 10])
def test_single_layer_kernels_with_exact_operands(dtype, layer):
    """wgrad / bias-grad / dgrad / forward of ONE layer on operands that are exact in bf16: the only error left is the
    fp32 accumulation order (and, for bf16 outputs, one final rounding)."""
    import ctypes
    import torch
    from speechless_amd import _lib
    from speechless_amd.engine import HALO
    rng = np.random.RandomState(40 + layer)
    case = make_case(b=3, t=150, seed=30)
    eng = make_engine(case, dtype)
    buf = eng.load_input(case["x"])
    buf.ensure_backward(eng)
    p = eng.plans[layer]
    s = p.spec
    t_out = buf.t_out
    td = eng.torch_dtype
    # ---- operands
    if layer == 0:
        x = _bf16_exact(rng, (3, 150, s.cin))
        eng.load_input(x)
        x_in = x
    else:
        x_in = np.maximum(_bf16_exact(rng, (3, t_out, s.cin)), 0)
        xt = torch.zeros_like(buf.y[layer - 1])
        xt[:, HALO:HALO + t_out, :s.cin] = torch.tensor(x_in).to(td)
        if eng._has_ones_output(eng.plans[layer - 1]):
            xt[:, HALO:HALO + t_out, -1] = 1  # the ones channel a forward pass would have put there
        buf.y[layer - 1].copy_(xt)
    g = _bf16_exact(rng, (3, t_out, s.cout), 0.01)
    gt = torch.zeros_like(buf.g[layer])
    gt[:, HALO:HALO + t_out, :s.cout] = torch.tensor(g).to(td)
    buf.g[layer].copy_(gt)
    w = _bf16_exact(rng, (s.kernel_size, s.cin, s.cout), 0.05)
    bias = _bf16_exact(rng, (s.cout,), 0.1)
    weights = list(case["weights"])
    weights[layer] = (w, bias)
    eng.set_weights(weights)
    eng.repack_weights()
    st = torch.cuda.current_stream().cuda_stream
    # ---- reference (float64)
    dx_ref, dw_ref, db_ref = o.conv1d_backward(x_in.astype(np.float64), w.astype(np.float64), s.stride,
                                               g.astype(np.float64))
    # ---- wgrad + bias grad
    dw_v, db_v = eng.layer_param_views(eng.grads, p)
    eng.lib.call("sl_conv1d_wgrad", (buf.x0 if layer == 0 else buf.y[layer - 1]).data_ptr(), buf.g[layer].data_ptr(),
                 dw_v.data_ptr(), ctypes.byref(buf.wgrad_geom[layer]), eng.dtype_code, 0, buf.wgrad_ws.data_ptr(),
                 buf.wgrad_ws.numel(), st)
    eng.lib.call("sl_bias_grad", buf.g[layer].data_ptr(), db_v.data_ptr(), ctypes.byref(buf.wgrad_geom[layer]),
                 eng.dtype_code, buf.bias_ws.data_ptr(), buf.bias_ws.numel(), st)
    torch.cuda.synchronize()
    dw = dw_v[:, :s.cin, :s.cout].cpu().numpy()
    db = db_v[:s.cout].cpu().numpy()
    assert rel_l2(dw, dw_ref) < 2e-6, ("wgrad", dtype, layer, rel_l2(dw, dw_ref))
    assert rel_l2(db, db_ref) < 2e-6, ("bias grad", dtype, layer, rel_l2(db, db_ref))
    full = dw_v.cpu().numpy()
    assert not full[:, :, s.cout:].any()  # padded lanes stay zero ...
    p = eng.plans[layer]
    if layer > 0 and eng._has_ones_output(eng.plans[layer - 1]):
        # ... except the row of the input's ones channel: at the tap that reads the frame itself it IS the bias gradient
        assert not full[:, s.cin:p.cin_pad - 1, :].any()
        assert rel_l2(full[p.pad_left, p.cin_pad - 1, :s.cout], db_ref) < 2e-6
    else:
        assert not full[:, s.cin:, :].any()
    # ---- dgrad (with the ReLU mask of the layer input)
    if layer > 0:
        eng.lib.call("sl_conv1d_nt", buf.g[layer].data_ptr(), eng.w_dgrad[layer].data_ptr(), None,
                     buf.y[layer - 1].data_ptr(), buf.g[layer - 1].data_ptr(), ctypes.byref(buf.dgrad_geom[layer]),
                     _lib.EPI_RELU_MASK, eng.dtype_code, 0, 0, buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
        torch.cuda.synchronize()
        raw = buf.g[layer - 1].float().cpu().numpy()
        got = raw[:, HALO:HALO + t_out, :s.cin]
        want = dx_ref * (x_in > 0)
        tol = 1e-5 if dtype == "f32" else 3e-3  # f32: sequential fp32 sum over up to 65536 terms; bf16: output rounding
        assert rel_l2(got, want) < tol, ("dgrad", dtype, layer, rel_l2(got, want))
        assert not raw[:, :HALO].any() and not raw[:, HALO + t_out:].any() and not raw[:, :, s.cin:].any()
    # ---- forward
    if layer < len(eng.plans) - 1:
        eng.lib.call("sl_conv1d_nt", (buf.x0 if layer == 0 else buf.y[layer - 1]).data_ptr(),
                     eng.w_fwd[layer].data_ptr(), eng.layer_param_views(eng.params, p)[1].data_ptr(), None,
                     buf.y[layer].data_ptr(), ctypes.byref(buf.fwd_geom[layer]), _lib.EPI_BIAS_RELU, eng.dtype_code, 0,
                     0, buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
        torch.cuda.synchronize()
        got = buf.y[layer].float().cpu().numpy()[:, HALO:HALO + t_out, :s.cout]
        want = np.maximum(o.conv1d_preactivation(x_in.astype(np.float64), w.astype(np.float64),
                                                 bias.astype(np.float64), s.stride), 0)
        tol = 1e-5 if dtype == "f32" else 3e-3
        assert rel_l2(got, want) < tol, ("forward", dtype, layer, rel_l2(got, want))


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_fused_adam_pack_equals_adam_then_pack(dtype):
    """sl_adam_pack_layer (one pass: Adam + both operand layouts) against sl_adam_step followed by sl_pack_weights, and
    the masters against the Keras-2.0 formula of the oracle."""
    import torch
    case = make_case(b=2, t=64, seed=8)
    engines = []
    for fused in (False, True):
        eng = make_engine(case, dtype, lr=1e-3)
        run_loss_and_grads(eng, case)
        for _ in range(2):  # two steps on the same gradient: exercises non-zero moments
            eng.adam_step(fused=fused)
            if not fused:
                eng.repack_weights()
        torch.cuda.synchronize()
        engines.append(eng)
    a, b = engines
    # not bitwise: hipcc contracts b1*m + (1-b1)*g into FMAs differently in the two kernels (last-ulp differences)
    for x, y in ((a.params, b.params), (a.adam_m, b.adam_m), (a.adam_v, b.adam_v)):
        assert torch.allclose(x, y, rtol=2e-6, atol=1e-9)
    ulp = 2.0 ** -7 if dtype == "bf16" else 1e-6  # a last-ulp master difference may flip one bf16 rounding
    for i in range(len(a.plans)):
        assert torch.allclose(a.w_fwd[i].float(), b.w_fwd[i].float(), rtol=ulp, atol=1e-9), i
        if a.w_dgrad[i] is not None:
            assert torch.allclose(a.w_dgrad[i].float(), b.w_dgrad[i].float(), rtol=ulp, atol=1e-9), i
        # and the two operand copies of the fused path hold exactly the rounded masters, in the right layout
        p = b.plans[i]
        wv, _ = b.layer_param_views(b.params, p)
        assert torch.equal(b.w_fwd[i], wv.permute(2, 0, 1).contiguous().to(b.torch_dtype)), i
        if b.w_dgrad[i] is not None:
            assert torch.equal(b.w_dgrad[i], wv.flip(0).permute(1, 0, 2).contiguous().to(b.torch_dtype)), i
    grads = a.get_gradients()
    for (w0, b0), (w2, b2), (gw, gb) in zip(case["weights"], a.get_weights(), grads):
        p, m, v = w0.astype(np.float64), np.zeros_like(w0, dtype=np.float64), np.zeros_like(w0, dtype=np.float64)
        pb, mb, vb = b0.astype(np.float64), np.zeros_like(b0, dtype=np.float64), np.zeros_like(b0, dtype=np.float64)
        for step in (1, 2):
            p, m, v = o.keras_adam_step(p, gw.astype(np.float64), m, v, step, lr=1e-3)
            pb, mb, vb = o.keras_adam_step(pb, gb.astype(np.float64), mb, vb, step, lr=1e-3)
        np.testing.assert_allclose(w2, p, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(b2, pb, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------ ELU, config-5 shapes
def test_elu_activation_forward_and_gradients_f32():
    """activation="elu" (used by the reference's main.py:71-78 runs): forward and all gradients vs the oracle."""
    from speechless_amd.engine import wav2letter_layer_specs
    case = make_case(b=2, t=64, seed=9)
    case["specs"] = wav2letter_layer_specs(128, 29, activation="elu")
    case["ospecs"] = o.layer_specs(128, 29, activation="elu")
    eng = make_engine(case, "f32")
    losses, grads = run_loss_and_grads(eng, case)
    ref = o.loss_and_gradients(case["ospecs"], weights64(case), case["x"].astype(np.float64), case["labels"],
                               case["prediction_lengths"], case["label_lengths"])
    np.testing.assert_allclose(losses, ref["losses"], rtol=1e-5)
    for i, ((dw, db), (rw, rb)) in enumerate(zip(grads, ref["grads"])):
        assert rel_l2(dw, rw) < 1e-4 and rel_l2(db, rb) < 1e-4, i
    engb = make_engine(case, "bf16")
    lb, _ = run_loss_and_grads(engb, case)
    np.testing.assert_allclose(lb, ref["losses"], rtol=2e-3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_power_spectrogram_shape_with_ragged_lengths(dtype):
    """BASELINE config 5 geometry at a size the oracle finishes in seconds: 257 input bins (padded to 320 lanes; pair
    view 640), ragged utterance lengths zero-padded to the batch maximum exactly like the reference (net.py:578-587),
    odd maximum length (T' = ceil(T/2), last frame never scored)."""
    rng = np.random.RandomState(14)
    lengths = [301, 180, 257, 96]
    x = np.zeros((4, max(lengths), 257), dtype=np.float32)
    for i, n in enumerate(lengths):
        x[i, :n] = rng.randn(n, 257)
    case = make_case(b=4, t=max(lengths), f=257, seed=15)
    case["x"] = x
    case["prediction_lengths"] = [n // 2 for n in lengths]
    case["label_lengths"] = [20, 9, 14, 5]
    case["labels"] = o.pack_label_batch([list(rng.randint(0, 28, size=n)) for n in case["label_lengths"]])
    eng = make_engine(case, dtype)
    losses, grads = run_loss_and_grads(eng, case)
    ref = o.loss_and_gradients(case["ospecs"], weights64(case), x.astype(np.float64), case["labels"],
                               case["prediction_lengths"], case["label_lengths"])
    if dtype == "f32":
        from speechless_amd.engine import HALO
        np.testing.assert_allclose(losses, ref["losses"], rtol=1e-5)
        # Back-propagated signal per layer and utterance.  A pre-activation within fp32 rounding of zero can take a
        # different sign in the fp32 forward than in the float64 oracle; that flips ONE ReLU-mask element (of ~3e5 per
        # utterance and layer) and moves that utterance's gradient by ~1/sqrt(n) = 2e-3 from there on (measured:
        # HISTORY.md section 1 "ReLU mask flips").  Any two float32 implementations differ like this, so: most utterances must
        # agree to 2e-4 at every layer, every one to 1e-2, and the weight gradients to 5e-3.
        for li in range(len(eng.plans)):
            s = eng.specs[li]
            g = eng.cur.g[li].float().cpu().numpy()[:, HALO:HALO + eng.cur.t_out, :s.cout]
            per_utt = sorted(rel_l2(g[b], ref["dzs"][li][b]) for b in range(g.shape[0]))
            assert per_utt[len(per_utt) // 2 - 1] < 2e-4 and per_utt[-1] < 1e-2, (s.name, per_utt)
        for i, ((dw, db), (rw, rb)) in enumerate(zip(grads, ref["grads"])):
            assert rel_l2(dw, rw) < 5e-3 and rel_l2(db, rb) < 5e-3, i
        decoded, _ = eng.greedy_decode(case["prediction_lengths"])
        assert decoded == o.greedy_decode_indices(ref["probs"], case["prediction_lengths"])
    else:
        np.testing.assert_allclose(losses, ref["losses"], rtol=2e-3)
        for name, i in (("output_conv", 10), ("big_conv_2", 9), ("big_conv_1", 8)):
            assert rel_l2(grads[i][0], ref["grads"][i][0]) < 3e-2, name


# ------------------------------------------------------------------------------------------ every NT tile configuration
def _nt_cfg(wm, wn, stages, ksplit=1, it=4, m32=0, gm_log2p1=0, slab=0):
    """cfg word of sl_conv1d_nt (see conv_nt_bf16.hip:decode_cfg); slab = 3 selects the interleaved slab schedule."""
    return (wm | (wn << 4) | (stages << 8) | (ksplit << 12) | (it << 20) | (m32 << 24) | (gm_log2p1 << 25) |
            ((slab & 1) << 29) | ((slab >> 1) << 30))


NT_CONFIGS = (
    # plain ring
    [(wm, wn, st, 1, it, 0, 0, 0) for (wm, wn, st, it) in
     [(2, 2, 2, 4), (2, 2, 4, 4), (4, 2, 3, 4), (2, 4, 2, 4), (4, 4, 2, 4), (2, 2, 3, 2), (4, 2, 4, 2), (8, 2, 2, 2),
      (2, 4, 2, 8)]] +
    # register-pipelined ring (stages | 8)
    [(wm, wn, st, 1, it, 0, 0, 0) for (wm, wn, st, it) in
     [(2, 2, 10, 4), (2, 2, 12, 4), (4, 2, 11, 2), (4, 2, 12, 2), (4, 2, 10, 4), (2, 4, 11, 4), (2, 4, 10, 8), (2, 2, 10, 8), (2, 2, 11, 8)]] +
    # 32x32x16 MFMA shape
    [(2, 2, 3, 1, 4, 1, 0, 0), (4, 4, 2, 1, 4, 1, 0, 0)] +
    # slab (chunk-major) variants, with and without split-K, and an explicit 2-D raster
    [(4, 4, 2, 1, 4, 0, 0, 1), (2, 4, 10, 1, 8, 0, 0, 1), (4, 2, 11, 1, 2, 0, 0, 1), (4, 2, 3, 1, 2, 0, 0, 1),
     (2, 2, 11, 1, 4, 0, 0, 1), (2, 2, 12, 1, 4, 0, 0, 1), (2, 4, 10, 2, 8, 0, 0, 1), (4, 2, 11, 4, 2, 0, 0, 1),
     (2, 4, 10, 1, 8, 0, 2, 1), (4, 4, 2, 1, 4, 0, 1, 0), (4, 2, 11, 1, 2, 0, 3, 0),
     (2, 4, 10, 1, 8, 0, 0, 3), (2, 4, 10, 2, 8, 0, 0, 3), (2, 4, 10, 4, 8, 0, 1, 3),
     # 128x128 tile with eight waves in k-half pairs (it = 5)
     (2, 2, 11, 1, 5, 0, 0, 2), (2, 2, 12, 1, 5, 0, 0, 2), (2, 2, 11, 2, 5, 0, 0, 2), (2, 2, 12, 3, 5, 0, 3, 2),
     # 4-wave 128x128 interleaved slab kernel, ring of 2 / 3 / 4 weight slots
     (2, 2, 10, 1, 4, 0, 0, 3), (2, 2, 11, 1, 4, 0, 0, 3), (2, 2, 12, 1, 4, 0, 0, 3), (2, 2, 11, 2, 4, 0, 0, 3), (2, 2, 12, 4, 4, 0, 2, 3),
     # interleaved tap-major (slab = 2: only the interleave bit)
     (4, 2, 10, 1, 2, 0, 0, 2), (4, 2, 11, 1, 2, 0, 0, 2), (4, 2, 12, 2, 2, 0, 0, 2), (2, 2, 10, 1, 4, 0, 0, 2),
     (2, 2, 11, 1, 4, 0, 0, 2), (2, 2, 12, 1, 4, 0, 0, 2), (2, 4, 10, 1, 8, 0, 0, 2)] +
    # split-K of the tap-major kernels
    [(4, 4, 2, 3, 4, 0, 0, 0), (4, 2, 11, 2, 2, 0, 0, 0)])


@pytest.mark.parametrize("taps,t_out,batch", [(7, 300, 3), (32, 140, 2), (5, 129, 5)])
def test_every_nt_tile_configuration_against_float64(hip_lib, taps, t_out, batch):
    """sl_conv1d_nt with EVERY instantiated tile configuration (plain / register-pipelined / slab / 32x32 MFMA /
    split-K / 2-D raster) on one geometry with bf16-exact operands, against a float64 evaluation of the same sum.
    Odd batches make the launch grid a non-multiple of 8 (padded grid + XCD remap)."""
    import ctypes
    import torch
    from speechless_amd import _lib
    rng = np.random.RandomState(taps * 1000 + t_out)
    cin, cout, halo = 256, 256, 48
    rows = halo + ((t_out + 255) // 256) * 256 + halo
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    x = np.zeros((batch, rows, cin), dtype=np.float32)
    x[:, halo:halo + t_out] = _bf16_exact(rng, (batch, t_out, cin))
    w = _bf16_exact(rng, (taps, cin, cout), 0.05)
    bias = _bf16_exact(rng, (cout,), 0.1)
    pad_l = (taps - 1) // 2
    xt = torch.tensor(x).to(torch.bfloat16).to(dev)
    wm_t = torch.tensor(w).to(dev)
    w_fwd = torch.zeros((cout, taps, cin), dtype=torch.bfloat16, device=dev)
    hip_lib.call("sl_pack_weights", wm_t.data_ptr(), w_fwd.data_ptr(), None, taps, cin, cout, _lib.SL_BF16, st)
    bias_t = torch.tensor(bias).to(dev)
    geom = _lib.ConvGeom()
    geom.batch, geom.t_out, geom.taps, geom.cin, geom.cout = batch, t_out, taps, cin, cout
    geom.x_row0, geom.x_row_stride, geom.x_batch_stride = halo - pad_l, cin, rows * cin
    geom.y_row0, geom.y_row_stride, geom.y_batch_stride = halo, cout, rows * cout
    # float64 reference: y[b,t,co] = bias + sum_{tap,ci} x[b, t + tap - pad_l, ci] w[tap, ci, co]
    xp = x.astype(np.float64)
    want = np.zeros((batch, t_out, cout))
    for tap in range(taps):
        want += xp[:, halo - pad_l + tap: halo - pad_l + tap + t_out] @ w[tap].astype(np.float64)
    want += bias
    ran = 0
    for (wm, wn, stg, ks, it, m32, gml, slab) in NT_CONFIGS:
        cfg = _nt_cfg(wm, wn, stg, ks, it, m32, gml, slab)
        need = hip_lib.raw("sl_conv1d_nt_workspace_bytes")(ctypes.byref(geom), _lib.SL_BF16, cfg)
        ws = torch.empty((max(int(need), 16),), dtype=torch.uint8, device=dev)
        y = torch.full((batch, rows, cout), 7.0, dtype=torch.float32, device=dev)
        rc = hip_lib.raw("sl_conv1d_nt")(xt.data_ptr(), w_fwd.data_ptr(), bias_t.data_ptr(), None, y.data_ptr(),
                                         ctypes.byref(geom), _lib.EPI_BIAS, _lib.SL_BF16, 1, cfg, ws.data_ptr(),
                                         ws.numel(), st)
        if rc != 0:  # a configuration the geometry rules out (taps < ring depth, split-K > chunks): must say so cleanly
            assert rc == -1 and "invalid tile configuration" in hip_lib.last_error(), (cfg, rc, hip_lib.last_error())
            continue
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        err = rel_l2(got[:, halo:halo + t_out], want)
        assert err < 2e-6, ("cfg", (wm, wn, stg, ks, it, m32, gml, slab), err)
        assert (got[:, :halo] == 7.0).all() and (got[:, halo + t_out:] == 7.0).all(), "rows outside [0, t_out) written"
        ran += 1
    assert ran >= len(NT_CONFIGS) - 8


WGRAD_CONFIGS = [(2, 2, 2), (2, 2, 3), (2, 2, 4), (4, 2, 2), (4, 2, 3), (2, 4, 2), (2, 4, 3), (4, 4, 2),
                 (4, 4, 10), (4, 4, 11)]  # stages 10: the 8-wave interleaved 256x256 kernel, 11: its 4-slot 32-frame ring


@pytest.mark.parametrize("taps,t_out,batch,groups", [(7, 200, 6, 1), (3, 130, 5, 3), (1, 64, 4, 1)])
def test_every_wgrad_tile_configuration_against_float64(hip_lib, taps, t_out, batch, groups):
    """sl_conv1d_wgrad / sl_conv1d_wgrad_grouped with every tile shape, ring depth and batch split (padded launch
    grids included) on bf16-exact operands against a float64 evaluation of the same sum; each configuration is run
    twice and must reproduce itself bit for bit (deterministic split reduction)."""
    import ctypes
    import torch
    from speechless_amd import _lib
    rng = np.random.RandomState(taps * 100 + batch)
    cin, cout, halo = 256, 256, 16
    rows = halo + ((t_out + 63) // 64) * 64 + halo
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    pad_l = (taps - 1) // 2
    x = np.zeros((groups, batch, rows, cin), dtype=np.float32)
    g = np.zeros((groups, batch, rows, cout), dtype=np.float32)
    x[:, :, halo:halo + t_out] = _bf16_exact(rng, (groups, batch, t_out, cin))
    g[:, :, halo:halo + t_out] = _bf16_exact(rng, (groups, batch, t_out, cout), 0.05)
    xt = torch.tensor(x).to(torch.bfloat16).to(dev)
    gt = torch.tensor(g).to(torch.bfloat16).to(dev)
    geom = _lib.ConvGeom()
    geom.batch, geom.t_out, geom.taps, geom.cin, geom.cout = batch, t_out, taps, cin, cout
    geom.x_row0, geom.x_row_stride, geom.x_batch_stride = halo - pad_l, cin, rows * cin
    geom.y_row0, geom.y_row_stride, geom.y_batch_stride = halo, cout, rows * cout
    want = np.zeros((groups, taps, cin, cout))
    for q in range(groups):
        for tap in range(taps):
            xs = x[q, :, halo - pad_l + tap: halo - pad_l + tap + t_out].astype(np.float64).reshape(-1, cin)
            want[q, tap] = xs.T @ g[q, :, halo:halo + t_out].astype(np.float64).reshape(-1, cout)
    dw_stride = taps * cin * cout + 64  # a gap between the groups' outputs, as in the engine (bias block)
    for (wm, wn, stg) in WGRAD_CONFIGS:
        for splits in (0, 1, 2, 3, batch):
            cfg = wm | (wn << 4) | (stg << 8) | (splits << 12)
            outs = []
            for _ in range(2):
                dw = torch.full((groups * dw_stride,), 3.0, dtype=torch.float32, device=dev)
                if groups == 1:
                    need = hip_lib.raw("sl_conv1d_wgrad_workspace_bytes")(ctypes.byref(geom), _lib.SL_BF16, cfg)
                    ws = torch.empty((max(int(need), 16),), dtype=torch.uint8, device=dev)
                    hip_lib.call("sl_conv1d_wgrad", xt.data_ptr(), gt.data_ptr(), dw.data_ptr(), ctypes.byref(geom),
                                 _lib.SL_BF16, cfg, ws.data_ptr(), ws.numel(), st)
                else:
                    need = hip_lib.raw("sl_conv1d_wgrad_grouped_workspace_bytes")(ctypes.byref(geom), groups, cfg)
                    ws = torch.empty((max(int(need), 16),), dtype=torch.uint8, device=dev)
                    hip_lib.call("sl_conv1d_wgrad_grouped", xt.data_ptr(), gt.data_ptr(), dw.data_ptr(),
                                 ctypes.byref(geom), groups, batch * rows * cin, batch * rows * cout, dw_stride, cfg,
                                 ws.data_ptr(), ws.numel(), st)
                torch.cuda.synchronize()
                outs.append(dw.cpu().numpy().reshape(groups, dw_stride))
            assert np.array_equal(outs[0], outs[1]), ("not deterministic", wm, wn, stg, splits)
            got = outs[0][:, :taps * cin * cout].reshape(groups, taps, cin, cout)
            err = rel_l2(got, want)
            assert err < 2e-6, ("cfg", (wm, wn, stg, splits), err)
            assert (outs[0][:, taps * cin * cout:] == 3.0).all(), "wrote past a group's weight block"


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("shard_optimizer", [False, True])
def test_data_parallel_step_through_rccl_single_rank(shard_optimizer, dtype):
    """The data-parallel step (bucketed exchange on the communication stream, overlapped with backward; with
    shard_optimizer: reduce-scatter, Adam on the rank's slice, all-gather of the masters, operand rewrite) on ONE rank
    through the real RCCL backend: a sum over a world of one is the identity, so weights and losses must equal the plain
    step bit for bit.  (World size 2 runs on CPU/gloo in tests/test_parallel.py; the 8-GPU run is the driver's.)"""
    import os
    import torch
    import torch.distributed as dist
    from speechless_amd.parallel import GradBucketReducer
    case = make_case(b=4, t=96, seed=5)
    results = []
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        for use_reducer in (False, True):
            eng = make_engine(case, dtype)
            reducer = None
            if use_reducer:
                reducer = GradBucketReducer(eng.grads, eng.bucket_ranges(), force=True, shard_optimizer=shard_optimizer)
            losses = []
            for _ in range(3):
                loss = eng.train_step(case["x"], case["labels"], np.array(case["label_lengths"]),
                                      np.array(case["prediction_lengths"]), reducer)
                losses.append(loss.cpu().numpy().copy())
            torch.cuda.synchronize()
            results.append((np.stack(losses), [w.copy() for w, _ in eng.get_weights()]))
    finally:
        if created:
            dist.destroy_process_group()
    assert np.array_equal(results[0][0], results[1][0])
    for a, b in zip(results[0][1], results[1][1]):
        assert np.array_equal(a, b)
    assert np.isfinite(results[0][0]).all() and (results[0][0][2] != results[0][0][0]).any()  # the weights did move


def test_prefetch_pipeline_trains_exactly_like_the_serial_loop(tmp_path):
    """train(prefetch_depth=2) -- worker-thread packing, H2D on a copy stream, slot recycling --
    must produce bit-identical weights to the reference's serial loop (prefetch_depth=0) on ragged batches."""
    from speechless_amd import Wav2Letter, english_frequent_characters
    from speechless_amd.net import Adam
    rng = np.random.RandomState(21)
    batches = [synthetic_examples(int(rng.randint(2, 6)), rng) for _ in range(14)]
    finals = []
    for depth in (0, 2):
        net = Wav2Letter(128, english_frequent_characters, optimizer=Adam(1e-3), seed=4)
        net.train(batches, preview_labeled_spectrogram_batch=batches[0][:2], tensor_board_log_directory=None,
                  net_directory=tmp_path / "n{}".format(depth), batches_per_epoch=4)
        finals.append([w.copy() for w, _ in net.predictive_net.get_weights()])
    for a, b in zip(*finals):
        assert np.array_equal(a, b)

    def failing():
        yield batches[0]
        raise RuntimeError("corpus reader died")
    net = Wav2Letter(128, english_frequent_characters, seed=4)
    with pytest.raises(RuntimeError, match="corpus reader died"):  # worker errors surface in the training loop
        net.train(failing(), preview_labeled_spectrogram_batch=batches[0][:2], tensor_board_log_directory=None,
                  net_directory=tmp_path / "x", batches_per_epoch=4)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_dropout_training_step_matches_the_oracle_with_the_same_masks(dtype):
    """Dropout (net.py:301-303) in front of the first n-3 layers: the masks the kernels drew are read back from the
    stored activations and handed to the oracle as explicit multipliers; loss and every gradient must then agree as
    tightly as without dropout.  Also: kept fraction ~ 1 - rate, a step is reproducible from its seed, inference is
    unaffected."""
    import torch
    case = make_case(b=3, t=96, seed=9)
    rate = 0.25
    eng = make_engine(case, dtype)
    probs_eval = eng.forward(case["x"]).cpu().numpy().copy()
    eng.dropout_rate = rate
    eng.dropout_seed = 11
    assert np.array_equal(eng.forward(case["x"]).cpu().numpy(), probs_eval)  # learning phase 0: no dropout
    eng.load_input(case["x"])
    eng.set_labels(case["labels"], np.array(case["label_lengths"]), np.array(case["prediction_lengths"]))
    eng.forward(training=True)
    loss = eng.ctc().cpu().numpy().copy()
    eng.backward()
    torch.cuda.synchronize()
    buf = eng.cur
    n = len(eng.plans)
    # ---- recover the multipliers: input of layer 0 from x0 / x0_dropped, input of layer i from the stored y[i-1]
    from speechless_amd.engine import HALO
    t_out = buf.t_out
    scales = [None] * n
    x0 = buf.x0.float().cpu().numpy()
    x0d = buf.x0_dropped.float().cpu().numpy()
    keep0 = (x0d != 0) | (x0 == 0)
    frac = keep0[x0 != 0].mean()
    assert abs(frac - (1 - rate)) < 0.01, frac
    # layer 0 reads the pair view of x0; the oracle wants the mask in (B, T, F) order: rebuild it by packing a ones tensor
    ones = np.ones_like(case["x"])
    eng2 = make_engine(case, dtype)
    eng2.load_input(ones)
    layout = eng2.cur.x0.float().cpu().numpy()  # 1 where a real input element sits
    idx = np.zeros_like(case["x"], dtype=np.int64)
    eng2.load_input(np.arange(1, case["x"].size + 1, dtype=np.float32).reshape(case["x"].shape) if dtype == "f32"
                    else ones)
    if dtype == "f32":
        pos = eng2.cur.x0.cpu().numpy()
        flat = np.zeros(case["x"].size + 1, dtype=bool)
        flat[pos[layout != 0].astype(np.int64)] = keep0[layout != 0]
        scales[0] = flat[1:].reshape(case["x"].shape) / (1 - rate)
    for i in range(1, n - 3):
        y = buf.y[i - 1].float().cpu().numpy()[:, HALO:HALO + t_out, :case["specs"][i].cin]
        scales[i] = ("from_output", y)
    if dtype != "f32":
        return  # bf16: statistics and plumbing only; the exact comparison runs in fp32
    # ---- oracle with explicit masks.  For layers >= 1 the multiplier is (stored y' != 0 or y == 0) / (1 - rate); where
    # the pre-dropout activation was 0 the choice does not matter, so derive it from the oracle's own forward pass.
    w64 = weights64(case)
    x64 = case["x"].astype(np.float64)
    input_scales = [scales[0].astype(np.float64)] + [None] * (n - 1)
    for i in range(1, n - 3):
        _, xs, zs = o.forward_stack(case["ospecs"], w64, x64, keep=True, input_scales=input_scales)
        stored = scales[i][1]
        input_scales[i] = np.where(stored != 0, 1.0 / (1 - rate), 0.0)
        pre = np.maximum(zs[i - 1], 0)
        assert rel_l2(stored, np.where(stored != 0, pre / (1 - rate), 0.0)) < 1e-5  # kept values are scaled up
    ref = o.loss_and_gradients(case["ospecs"], w64, x64, case["labels"], case["prediction_lengths"],
                               case["label_lengths"], input_scales=input_scales)
    assert np.allclose(loss, ref["losses"], rtol=2e-5)
    errs = [max(rel_l2(dw, rw), rel_l2(db, rb)) for (dw, db), (rw, rb) in zip(eng.get_gradients(), ref["grads"])]
    # flip-aware (HISTORY.md section 1 "ReLU mask flips"): one pre-activation within fp32 rounding of zero that takes the other
    # branch than in float64 moves the gradients of all layers BELOW it by ~1/sqrt(elements) = 2e-3 here
    loose = [i for i, e in enumerate(errs) if e >= 2e-4]
    assert max(errs) < 1e-2 and loose == list(range(len(loose))) and errs[-1] < 2e-4, errs
    # ---- reproducible from the seed, different with another seed
    eng3 = make_engine(case, dtype)
    eng3.dropout_rate, eng3.dropout_seed = rate, 11
    eng3.load_input(case["x"])
    eng3.set_labels(case["labels"], np.array(case["label_lengths"]), np.array(case["prediction_lengths"]))
    eng3.forward(training=True)
    assert np.array_equal(eng3.ctc().cpu().numpy(), loss)
    eng3.dropout_seed = 12
    eng3._dropout_steps = 0
    eng3.forward(training=True)
    assert not np.array_equal(eng3.ctc().cpu().numpy(), loss)


def test_wav2letter_with_dropout_trains_and_predicts_without_it(tmp_path):
    from speechless_amd import Wav2Letter, english_frequent_characters
    from speechless_amd.net import Adam
    rng = np.random.RandomState(17)
    batch = synthetic_examples(4, rng)
    net = Wav2Letter(128, english_frequent_characters, optimizer=Adam(1e-3), dropout=0.1, seed=2)
    before = net.test_and_predict_batch(batch)
    assert before.average_loss == net.test_and_predict_batch(batch).average_loss  # evaluation is deterministic
    for _ in range(12):
        net.train_on_batch(batch)
    assert net.test_and_predict_batch(batch).average_loss < before.average_loss
    with pytest.raises(ValueError):
        Wav2Letter(128, english_frequent_characters, dropout=1.0)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_training_overfits_a_small_batch_to_exact_transcripts(dtype):
    """End-to-end evidence that forward, CTC, backward and Adam fit together over many steps: at the reference's
    learning rate (Adam(1e-4), net.py:132) four noise 'utterances' with four-word transcripts are memorised -- loss ~ 0
    and the greedy decode reproduces every transcript -- on the bf16 path as well as on the fp32 one."""
    from speechless_amd import Wav2Letter, english_frequent_characters
    from speechless_amd.net import Adam, LabeledSpectrogram
    rng = np.random.RandomState(3)
    words = ["she", "had", "your", "dark", "suit", "in", "greasy", "wash", "water", "all", "year"]
    batch = [LabeledSpectrogram(id=str(i), label=" ".join(rng.choice(words, size=4)),
                                spectrogram=rng.randn(160 + 10 * i, 128)) for i in range(4)]
    net = Wav2Letter(128, english_frequent_characters, optimizer=Adam(1e-4), seed=1, compute_dtype=dtype)
    first = net.train_on_batch(batch)
    for _ in range(299):
        net.train_on_batch(batch, lazy=True)
    result = net.test_and_predict_batch(batch)
    assert first > 100 and result.average_loss < 0.05
    assert [r.predicted for r in result.results] == [r.expected for r in result.results]
    assert result.average_letter_error_rate == 0.0


@pytest.mark.parametrize("b,t,f,k,sizes", [
    (1, 31, 128, 29, None),                                                       # a batch of ONE, odd length
    (2, 2, 128, 29, None),                                                        # two input frames -> one output frame
    (3, 70, 128, 33, None),                                                       # German alphabet: 32 graphemes + blank
    (2, 90, 40, 29, dict(main_filter_count=100, out_filter_count=300)),           # channel counts far from the padding grid
    (2, 66, 257, 29, dict(striding_kernel=12, inner_kernel=3, big_kernel=5, inner_count=2)),  # other kernel sizes / depth
])
def test_loss_and_gradients_f32_unusual_shapes(b, t, f, k, sizes):
    """The fp32 path against the oracle on shapes off the beaten track: batch 1, a single output frame, K = 33, channel
    counts that need heavy padding, other kernel sizes and a shallower stack (the kernels are generic in all of them)."""
    case = make_case(b=b, t=t, f=f, k=k, seed=23, sizes=sizes)
    eng = make_engine(case, "f32")
    losses, grads = run_loss_and_grads(eng, case)
    ref = o.loss_and_gradients(case["ospecs"], weights64(case), case["x"].astype(np.float64), case["labels"],
                               case["prediction_lengths"], case["label_lengths"])
    finite = np.isfinite(ref["losses"])
    np.testing.assert_allclose(losses[finite], ref["losses"][finite], rtol=2e-5)
    assert np.array_equal(np.isfinite(losses), finite)
    if finite.all():
        for i, ((dw, db), (rw, rb)) in enumerate(zip(grads, ref["grads"])):
            assert rel_l2(dw, rw) < 2e-4 and rel_l2(db, rb) < 2e-4, (i, rel_l2(dw, rw), rel_l2(db, rb))
    # the bf16 path runs the same shapes (tile tables, padding, grid rounding) and must stay close to the fp32 one
    eng16 = make_engine(case, "bf16")
    losses16, _ = run_loss_and_grads(eng16, case)
    np.testing.assert_allclose(losses16[finite], ref["losses"][finite], rtol=2e-2)


def test_full_length_loss_and_gradients_against_the_cpu_path():
    """BASELINE config 2/3 shapes (128 mel x 1000 frames, labels up to 200 graphemes), 4 utterances, directly against
    the CPU path (torch-CPU fp32 realisation of the oracle, itself pinned to the numpy restatement in
    tests/test_oracle.py): per-utterance CTC loss within 1e-3 relative on the bf16 path (north_star's bar) and 2e-5 on
    the fp32 path; fp32 gradients within 5e-4 rel-L2 for output_conv (1e-2 below: ReLU flips between two fp32
    summation orders); probabilities and greedy decode of the fp32 path exact to 2e-5 / bit-exact."""
    from oracle import w2l_torch_cpu as tc
    case = make_case(b=4, t=1000, seed=41)
    rng = np.random.RandomState(41)
    lab_len = [200, 137, 20, 75]
    labels = o.pack_label_batch([list(rng.randint(0, case["k"] - 1, size=n)) for n in lab_len])
    pred_len = [500, 500, 480, 500]
    ref = tc.loss_and_gradients(case["ospecs"], case["weights"], case["x"], labels, pred_len, lab_len)
    for dtype, loss_tol in (("f32", 2e-5), ("bf16", 1e-3)):
        eng = make_engine(case, dtype)
        eng.load_input(case["x"])
        eng.set_labels(labels, np.array(lab_len), np.array(pred_len))
        probs = eng.forward().cpu().numpy()
        losses = eng.ctc().cpu().numpy()
        eng.backward()
        np.testing.assert_allclose(losses, ref["losses"], rtol=loss_tol)
        _report("full_length_loss_rel_err_vs_cpu_{}".format(dtype), float(np.abs(losses / ref["losses"] - 1).max()))
        if dtype == "f32":
            decoded, _ = eng.greedy_decode(pred_len)
            assert decoded == o.greedy_decode_indices(ref["probs"], pred_len)
            assert np.abs(probs - ref["probs"]).max() < 2e-5
            # two fp32 implementations with different summation orders: a pre-activation within rounding of zero takes
            # the other ReLU branch now and then (~2e-3 of an utterance's gradient per flip, see the config-5 test), which
            # the layers below inherit; only output_conv sits above every ReLU that can flip
            grads = eng.get_gradients()
            n = len(grads)
            for i, ((dw, db), (rw, rb)) in enumerate(zip(grads, ref["grads"])):
                tol = 5e-4 if i == n - 1 else 1e-2  # output_conv: fp32 CTC over 500 frames + a 16000-term fp32 sum
                assert rel_l2(dw, rw) < tol and rel_l2(db, rb) < tol, (i, rel_l2(dw, rw), rel_l2(db, rb))
