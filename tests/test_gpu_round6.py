"""GPU parity tests added in round 6 (run with -m gpu on an MI355X).  Helpers come from the earlier GPU test modules."""
import numpy as np
import pytest

from oracle import w2l_oracle as o
from test_gpu_parity import _report, make_case, make_engine, rel_l2, weights64

pytestmark = pytest.mark.gpu


class _Example:
    """LabeledSpectrogram duck type (labeled_example.py:63-71)"""

    def __init__(self, spectrogram, label, id="x"):
        self._s, self.label, self.id = spectrogram, label, id

    def z_normalized_transposed_spectrogram(self):
        return self._s


# ------------------------------------------------------------------------------------------ the default path IS a parity path
def test_reference_signature_decodes_config2_bit_exactly_against_the_cpu_port():
    """VERDICT r5 item 2.  `Wav2Letter(128, english_frequent_characters)` -- the REFERENCE'S signature, nothing else, as
    configuration.py:106,168 calls it -- decodes BASELINE config 2 (32 x 128 mel x 1000 frames, random init) bit-exactly
    against the torch-CPU fp32 port of the reference's arithmetic: every forward-only entry point (prediction_batch,
    predict_batch_greedily, test_and_predict_batch; net.py:350-357, 461-498) runs the bf16x3 parity engine over the fp32
    masters the bf16 training engine updates (f16x3 since the fp16 planes exist; bf16x3 for raw-wave input or out-of-range weights).  Also: the per-utterance losses of test_and_predict_batch against the CPU
    port (1e-5), and that evaluation follows a training step (the shared masters are re-packed, not a stale copy)."""
    import torch
    from oracle import w2l_torch_cpu as tc
    from speechless_amd import Wav2Letter, english_frequent_characters
    net = Wav2Letter(128, english_frequent_characters)
    assert net.compute_dtype == "bf16" and net.eval_dtype == "f16x3"    # training: the benchmarked path; evaluation: parity
    rng = np.random.RandomState(0)
    x = rng.randn(32, 1000, 128).astype(np.float32)
    ospecs = o.layer_specs(128, 29)

    def cpu_probs():
        with torch.no_grad():
            w = tc.to_torch_weights(net.predictive_net.get_weights(), requires_grad=False)
            return tc.forward_probs(ospecs, w, torch.from_numpy(x)).numpy()

    ref = cpu_probs()
    want = [net.grapheme_encoding.decode_graphemes(d, merge_repeated=False)
            for d in o.greedy_decode_indices(ref, [500] * 32)]
    probs = net.prediction_batch(x)
    _report("default_path_config2_max_abs_prob_error", float(np.abs(probs - ref).max()))
    assert np.abs(probs - ref).max() < 2e-5
    assert np.array_equal(probs.argmax(axis=2), ref.argmax(axis=2))
    assert net.predict_batch_greedily(list(x)) == want
    # test_and_predict_batch: decode + loss from ONE forward of the parity engine
    lrng = np.random.RandomState(1)
    labels = ["".join(lrng.choice(list(english_frequent_characters), size=int(lrng.randint(20, 201)))) for _ in range(32)]
    labels = [" ".join(l.split()) or "a" for l in labels]  # (the codec needs no normalisation; keep labels non-empty)
    batch = [_Example(x[i], labels[i]) for i in range(32)]
    result = net.test_and_predict_batch(batch)
    assert [r.predicted for r in result.results] == want
    enc = net.grapheme_encoding.encode_label_batch(labels)
    lab_len = [len(l) for l in labels]
    ref_loss, _ = o.ctc_batch_cost(ref.astype(np.float64), enc, [500] * 32, lab_len)
    np.testing.assert_allclose([r.loss for r in result.results], ref_loss, rtol=2e-5)
    # a training step on the bf16 engine moves the masters; the next evaluation must see them
    before = net.predictive_net.get_weights()[3][0].copy()
    net.engine.lr = 1e-3
    net.train_on_batch(batch)
    after = net.predictive_net.get_weights()[3][0]
    assert not np.array_equal(before, after)
    ref2 = cpu_probs()
    probs2 = net.prediction_batch(x)
    assert np.abs(probs2 - ref2).max() < 2e-5 and np.abs(probs2 - ref).max() > 1e-6
    assert np.array_equal(probs2.argmax(axis=2), ref2.argmax(axis=2))
    # an explicit dtype keeps everything on that one engine (the benchmark's and the earlier tests' behaviour)
    one = Wav2Letter(128, english_frequent_characters, compute_dtype="bf16", seed=3)
    assert one.eval_engine is one.engine


def test_eval_engine_shares_the_masters_and_has_no_optimizer_state():
    from speechless_amd import Wav2Letter, english_frequent_characters
    small = dict(main_filter_count=20, out_filter_count=40, inner_count=1)
    net = Wav2Letter(128, english_frequent_characters, seed=4, layer_sizes=small)
    ev = net.eval_engine
    assert ev is not net.engine and ev.dtype == "f16x3" and ev.forward_only
    assert ev.params.data_ptr() == net.engine.params.data_ptr() and ev.grads is None and ev.adam_m is None
    with pytest.raises(RuntimeError):
        ev.adam_step()
    # set_weights through the public handle reaches the evaluation engine
    x = np.random.RandomState(0).randn(2, 64, 128).astype(np.float32)
    p0 = net.prediction_batch(x)
    w = net.predictive_net.get_weights()
    w[-1] = (w[-1][0] * 0.5, w[-1][1] + 0.25)
    net.predictive_net.set_weights(w)
    p1 = net.prediction_batch(x)
    assert np.abs(p0 - p1).max() > 1e-4
    ref, _, _ = o.forward_stack(o.layer_specs(128, 29, **small), [(a.astype(np.float64), b.astype(np.float64)) for a, b in w],
                                x.astype(np.float64), keep=True)
    assert np.abs(p1 - ref).max() < 2e-5
    # weights beyond what fp16 planes hold (w_scale * |w| < 65504): the evaluation engine becomes the bf16x3 one, results stay right
    w[-1][0][0, 3, 5] = 1500.0   # (one output-layer weight: the logits stay moderate, the range check must still fire)
    net.predictive_net.set_weights(w)
    p2 = net.prediction_batch(x)
    assert net.eval_dtype == "bf16x3" and net.eval_engine.dtype == "bf16x3" and np.isfinite(p2).all()
    ref2, _, _ = o.forward_stack(o.layer_specs(128, 29, **small), [(a.astype(np.float64), b.astype(np.float64)) for a, b in w],
                                 x.astype(np.float64), keep=True)
    assert np.abs(p2 - ref2).max() < 1e-3
    # raw-wave nets evaluate on bf16x3 from the start (the front layer has no fp16-plane form)
    assert Wav2Letter(1, english_frequent_characters, use_raw_wave_input=True, seed=1,
                      layer_sizes=dict(main_filter_count=250, out_filter_count=256, inner_count=1)).eval_dtype == "bf16x3"


# ------------------------------------------------------------------------------------------ stride-1 first layer on bf16x3
@pytest.mark.parametrize("f", [128, 64])
def test_bf16x3_with_a_first_layer_that_does_not_stride(f):
    """the five-plane / two-plane window geometry of the striding layer's weight gradient belongs to the
    PAIR VIEW (stride 2) only.  A stack whose first layer has stride 1 (engine.py accepts it) must get the ordinary
    [hi | lo] x g_hi + hi x g_lo launches: with 128 bins the round-5 geometry asked for 5 x 128 input columns out of rows of
    3 x 128 and the backward pass raised; with 64 it ran and contracted a plane that is not part of the product."""
    import torch
    sizes = dict(main_filter_count=250, out_filter_count=256, inner_count=2, striding_kernel=5)
    case = make_case(b=3, t=96, f=f, seed=7, sizes=sizes)
    for specs in (case["specs"], case["ospecs"]):
        specs[0].stride = 1
    pred_len = [96, 90, 77]
    eng = make_engine(case, "bf16x3")
    eng.load_input(case["x"])
    eng.set_labels(case["labels"], np.array(case["label_lengths"]), np.array(pred_len))
    eng.forward()
    losses = eng.ctc().cpu().numpy()
    eng.backward()
    torch.cuda.synchronize()
    ref = o.loss_and_gradients(case["ospecs"], weights64(case), case["x"].astype(np.float64), case["labels"], pred_len,
                               case["label_lengths"])
    np.testing.assert_allclose(losses, ref["losses"], rtol=2e-5)
    errs = {s.name: [rel_l2(dw, rw), rel_l2(db, rb)] for s, (dw, db), (rw, rb) in
            zip(case["specs"], eng.get_gradients(), ref["grads"])}
    _report("bf16x3_stride1_first_layer_f{}".format(f), errs)
    for name, (ew, eb) in errs.items():
        # (the first layer's dW is the tensor the ReLU decisions that differ from float64 show in: DESIGN.md section 1)
        assert ew < (1e-2 if name == "striding_conv" else 2e-3) and eb < 2e-3, (name, ew, eb)


# ------------------------------------------------------------------------------------------ data parallel: the last bucket's split
def test_split_last_bucket_plan_and_step():
    """Engine.split_last_bucket (off by default): the run of inner layers is cut at inner_conv_4 -- bucket plan
    {output_conv, big_conv_2}, {big_conv_1}, {inner_conv_4..7}, {striding_conv, inner_conv_1..3}, contiguous ranges that cover
    what the unsplit plan covers; the buckets are announced in that order, once each; the step's gradients equal the unsplit
    step's up to the summation order of the balanced launch (two launches partition their tiles differently), the fused
    input-gradient launch cut in two is bit-identical (activation gradients are, layer by layer)."""
    import torch
    case = make_case(b=4, t=300, seed=12)
    results = {}
    for split in (False, True):
        eng = make_engine(case, "bf16")
        eng.split_last_bucket = split
        plan = eng.bucket_plan()
        announced = []
        eng.load_input(case["x"])
        eng.set_labels(case["labels"], np.array(case["label_lengths"]), np.array(case["prediction_lengths"]))
        for _ in range(3):  # eager + recorded + replayed
            eng.forward(training=True)
            eng.ctc()
            eng.backward(on_bucket_ready=announced.append)
        torch.cuda.synchronize()
        results[split] = (plan, announced, eng.get_gradients(), [eng.cur.g[i].float().cpu().numpy() for i in range(8)])
    plan0, ann0, g0, a0 = results[False]
    plan1, ann1, g1, a1 = results[True]
    assert [layers for layers, _ in plan0] == [[9, 10], [8], [0, 1, 2, 3, 4, 5, 6, 7]]
    assert [layers for layers, _ in plan1] == [[9, 10], [8], [4, 5, 6, 7], [0, 1, 2, 3]]
    assert ann0 == [0, 1, 2] * 3 and ann1 == [0, 1, 2, 3] * 3
    (lo_a, hi_a), (lo_b, hi_b) = plan1[2][1], plan1[3][1]
    assert (lo_b, hi_a) == plan0[2][1] and hi_b == lo_a         # two contiguous halves of the unsplit bucket
    for x, y in zip(a0, a1):
        assert np.array_equal(x, y)                             # back-propagated signal: bit-identical
    for i, ((dw0, db0), (dw1, db1)) in enumerate(zip(g0, g1)):
        if i >= 8:
            assert np.array_equal(dw0, dw1) and np.array_equal(db0, db1)
        else:
            assert rel_l2(dw1, dw0) < 2e-6 and rel_l2(db1, db0) < 2e-6, (i, rel_l2(dw1, dw0))


def test_first_dp_run_matrix_control_flow(tmp_path):
    """tools/first_dp_run.sh -- the one command of the first real N > 1 run -- with eight ranks sharing the test box's one GPU
    over gloo (SL_BENCH_SHARE_GPU=1: control flow only, never a measurement): every combination of exchange (all-reduce /
    sharded optimizer) x CU hint (0 / 32) x last-bucket split (0 / 1) and one RCCL-environment variant runs to its JSON line
    with identical weights on all ranks, and the table names them all."""
    import json
    import os
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    out = tmp_path / "dp"
    env = dict(os.environ, SL_BENCH_SHARE_GPU="1", STEPS="2", WARMUP="1", CONFIGS="3",
               RCCL_VARIANTS="default| ring|NCCL_ALGO=Ring")
    res = subprocess.run(["bash", str(root / "tools" / "first_dp_run.sh"), "8", str(out)], capture_output=True, text=True,
                         env=env, cwd=str(root), timeout=3000)
    assert res.returncode == 0, res.stderr[-2000:]
    status = (out / "status.txt").read_text().split()
    assert status[0::2] == ["0"] * 9, (out / "status.txt").read_text()
    tags = status[1::2]
    assert len(set(tags)) == 9 and "c3_shard_cus32_split1_default" in tags and "c3_allreduce_cus0_split0_ring" in tags
    for tag in tags:
        line = json.loads([l for l in (out / (tag + ".json")).read_text().splitlines() if l.startswith("{")][-1])
        dp = line["data_parallel"]
        assert line["n_gpus"] == 8 and dp["reduced_gradients_and_weights_identical_on_all_ranks"] is True, tag
        opts = dp["options"]
        assert opts["split_last_bucket"] == ("split1" in tag) and opts["comm_cus"] == (32 if "cus32" in tag else 0), tag
        assert dp["sharded_optimizer"] == ("shard" in tag) and len(dp["bucket_bytes"]) == (4 if "split1" in tag else 3), tag
    table = (out / "table.txt").read_text()
    assert "FAILED" not in table and table.count("identical True") == 9 and "single GPU" in table


# ------------------------------------------------------------------------------------------ f16x3: the plane scheme on fp16 pairs
@pytest.mark.parametrize("t", [64, 77])
def test_f16x3_loss_and_gradients_against_the_float64_oracle(t):
    """Engine(dtype='f16x3'): every value as hi + lo FP16 planes -- 22 significand bits instead of bf16x3's
    16-17, same three MFMA terms (v_mfma_f32_16x16x32_f16), weights stored x 2^6 and gradients x 2^12 (exact powers of two,
    divided out by sl_conv_geom.acc_scale / the combine) -- against the float64 oracle: loss and every gradient tensor at the
    exact-fp32 path's level: north_star's 1e-3 on every tensor (2e-3 on striding_conv's dW, where a single ReLU decision that
    differs from float64 shows in a batch of three short utterances); bf16x3's bounds in tests/test_gpu_round3.py are 5e-3."""
    import torch
    case = make_case(b=3, t=t, seed=3)
    ref = o.loss_and_gradients(case["ospecs"], weights64(case), case["x"].astype(np.float64), case["labels"],
                               case["prediction_lengths"], case["label_lengths"])
    eng = make_engine(case, "f16x3")
    eng.load_input(case["x"])
    eng.set_labels(case["labels"], np.array(case["label_lengths"]), np.array(case["prediction_lengths"]))
    eng.forward()
    losses = eng.ctc().cpu().numpy()
    eng.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses, ref["losses"], rtol=1e-5)
    report = {}
    for spec, (dw, db), (rw, rb) in zip(case["specs"], eng.get_gradients(), ref["grads"]):
        report[spec.name] = [rel_l2(dw, rw), rel_l2(db, rb)]
    _report("grads_f16x3_rel_l2_t{}".format(t), report)
    for name, (ew, eb) in report.items():
        # (striding_conv's dW: where a single ReLU decision that differs from float64 shows)
        assert ew < (2e-3 if name == "striding_conv" else 1e-3) and eb < 1e-3, (name, ew, eb)  # north_star's bar
    decoded, _ = eng.greedy_decode(case["prediction_lengths"])
    assert decoded == o.greedy_decode_indices(ref["probs"], case["prediction_lengths"])
    # optimisation steps through the scaled plane operands: the loss goes down, padded lanes stay zero, and the operand copies
    # ARE w_scale * w in two fp16 planes
    for _ in range(6):
        eng.train_step_resident()
    torch.cuda.synchronize()
    assert float(eng.cur.loss.mean().item()) < float(np.mean(losses))
    p1 = eng.plans[1]
    full = eng.layer_param_views(eng.params, p1)[0]
    assert not full[:, 250:255, :].any() and not full[:, :, 250:].any()
    wf = eng.w_fwd[1].float()                                       # [cout][k][3 cin]: [w_hi | w_hi | w_lo]
    c = p1.cin_pad
    recon = (wf[:, :, :c] + wf[:, :, 2 * c:]) / eng.w_scale
    want = full.permute(2, 0, 1)
    assert torch.equal(wf[:, :, :c], wf[:, :, c:2 * c])
    assert float((recon - want).abs().max()) <= 2.0 ** -22 * float(want.abs().max())


def test_f16x3_config2_greedy_decode_bit_exact_at_batch_32():
    """BASELINE config 2 on the f16x3 path: frame argmax and decoded indices bit-exact against the torch-CPU fp32 port,
    probabilities within 2e-6 (bf16x3: 2e-5)."""
    import torch
    from oracle import w2l_torch_cpu as tc
    case = make_case(b=32, t=1000, seed=2)
    pred_len = [500] * 32
    with torch.no_grad():
        ref_probs = tc.forward_probs(case["ospecs"], tc.to_torch_weights(case["weights"], requires_grad=False),
                                     torch.from_numpy(case["x"])).numpy()
    eng = make_engine(case, "f16x3")
    probs = eng.forward(case["x"]).cpu().numpy()
    decoded, frame_argmax = eng.greedy_decode(pred_len)
    _report("config2_b32_f16x3_max_abs_prob_error", float(np.abs(probs - ref_probs).max()))
    assert np.abs(probs - ref_probs).max() < 2e-6
    assert np.array_equal(frame_argmax, ref_probs.argmax(axis=2))
    assert decoded == o.greedy_decode_indices(ref_probs, pred_len)


@pytest.mark.parametrize("activation", ["relu", "elu"])
def test_f16x3_dropout_and_elu_against_the_f32_path(activation):
    """f16x3 through its other helpers: ELU layers (fp32 staging + sl_splitf16 modes 2 / 4) and dropout on fp16 planes
    (sl_splitf16_dropout: the same (seed, element) keep decisions as every other path) -- one training step against the
    exact-fp32 path with the same seed."""
    import torch
    sizes = dict(main_filter_count=250, out_filter_count=256, inner_count=2)
    case = make_case(b=3, t=90, seed=21, sizes=sizes)
    for specs in (case["specs"], case["ospecs"]):
        for sp in specs[:-1]:
            sp.activation = activation
    out = {}
    for dtype in ("f16x3", "f32"):
        eng = make_engine(case, dtype)
        eng.dropout_rate, eng.dropout_seed = 0.2, 77
        eng.load_input(case["x"])
        eng.set_labels(case["labels"], np.array(case["label_lengths"]), np.array(case["prediction_lengths"]))
        eng.forward(training=True)
        losses = eng.ctc().cpu().numpy().copy()
        eng.backward()
        torch.cuda.synchronize()
        out[dtype] = (losses, eng.get_gradients())
    np.testing.assert_allclose(out["f16x3"][0], out["f32"][0], rtol=2e-5)
    errs = [max(rel_l2(a, c), rel_l2(b, d)) for (a, b), (c, d) in zip(out["f16x3"][1], out["f32"][1])]
    _report("f16x3_dropout_{}_vs_f32".format(activation), errs)
    assert max(errs) < (2e-4 if activation == "elu" else 5e-3), errs   # (ReLU: a flipped decision next to a dropout mask)


# ------------------------------------------------------------------------------------------ CTC: the lane that missed its lift
def test_ctc_tight_alignment_behind_a_blank_collapse_needs_no_repair(hip_lib):
    """Found by `tools/fuzz_ctc.py --seed 64` (case 136) at the end of round 6: 250 labels in 299 frames, the first half a learnt
    alignment, the second a blank collapse.  The probability-domain lattice lifts lanes that hold stale mass thousands of binades
    below the approaching front, five lanes per rescale; the lane one further on was skipped, and at the next rescale its source
    neighbour -- flushed to zero by its own lift -- no longer counted as holding mass: beta's front crossed into lane 0 under the
    clamp of the exponent difference and was crushed (posteriors of frames 0..3 zero; the utterance went through the repair pass:
    correct to the log-domain lattice's 6e-4 instead of 2e-6, at 0.7 ms per call).  With the lift also applied against a source
    neighbour that is zero but still in play (ctc.hip, rescale) the lattice needs no repair: results with and without the repair
    pass are bit-identical and within 1e-5 of the float64 oracle."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    from fuzz_ctc import replay_case
    from test_gpu_parity import run_ctc_kernel
    k, t, input_len, lab_len, labels_list, logits, desc = replay_case(64, 136)
    assert lab_len[1] == 250 and input_len[1] == 299 and desc[1] == "learnt|collapse"
    labels = o.pack_label_batch([l if l else [-1] for l in labels_list])
    ref_p = o.softmax(logits.astype(np.float64))
    ref_loss, ref_dp = o.ctc_batch_cost(ref_p, labels, input_len, lab_len)
    ref_dl = o.softmax_backward(ref_p, ref_dp)
    results = {}
    try:
        for variant in (0, 11, 4, 2):   # default (helper wave + repair), helper wave alone, lone wave + repair, lone wave alone
            hip_lib.call("sl_ctc_select", variant)
            _, loss, dl = run_ctc_kernel(hip_lib, logits, labels, lab_len, input_len)
            results[variant] = (loss, dl)
    finally:
        hip_lib.call("sl_ctc_select", 0)
    loss, dl = results[0]
    for variant in (11, 4, 2):
        assert np.array_equal(loss, results[variant][0]) and np.array_equal(dl, results[variant][1]), variant
    np.testing.assert_allclose(loss, ref_loss, rtol=2e-6)
    assert np.abs(dl - ref_dl).max() < 1e-5


@pytest.mark.parametrize("seed,index,tight", [(72, 60, False), (80, 160, True)])
def test_ctc_repair_pass_is_as_accurate_as_the_lattice_it_replaces(hip_lib, seed, index, tight):
    """Two more cases of the end-of-round soak in which the probability-domain lattice loses mass (alignments with next to no
    slack: 178 labels in 192 frames; 255 in 284) and the gradient kernel's posterior check sends the utterance through the
    repair pass.  Until round 6 that pass was the fp32 log-domain lattice -- log values of magnitude 2^12 carry 2.4e-4 each --
    and a repaired utterance came back at 7e-5 ... 2e-3 of the float64 oracle; it now runs in doubles (repair_lattices,
    ctc_grad_frames<., 4>).  Default variant (repairs what is flagged) and variant 3 (repairs EVERY utterance): loss to 2e-6,
    gradient to 1e-5 absolute."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import fuzz_ctc
    from test_gpu_parity import run_ctc_kernel
    rng = np.random.RandomState(seed)
    for _ in range(index + 1):
        k, t, input_len, lab_len, labels_list, logits, desc = fuzz_ctc.draw_case(rng, tight=tight)
    labels = o.pack_label_batch([l if l else [-1] for l in labels_list])
    ref_p = o.softmax(logits.astype(np.float64))
    ref_loss, ref_dp = o.ctc_batch_cost(ref_p, labels, input_len, lab_len)
    ref_dl = o.softmax_backward(ref_p, ref_dp)
    fin = np.isfinite(ref_loss)
    try:
        for variant in (0, 3):
            hip_lib.call("sl_ctc_select", variant)
            _, loss, dl = run_ctc_kernel(hip_lib, logits, labels, lab_len, input_len)
            assert np.array_equal(np.isinf(loss), ~fin)
            assert np.all(np.abs(loss[fin] - ref_loss[fin]) < 2e-6 * np.maximum(np.abs(ref_loss[fin]), 20.0)), (variant, loss, ref_loss)
            assert np.abs(dl[fin] - ref_dl[fin]).max() < 1e-5, (variant, np.abs(dl[fin] - ref_dl[fin]).max())
    finally:
        hip_lib.call("sl_ctc_select", 0)
