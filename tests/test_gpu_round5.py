"""GPU parity tests added in round 5 (run with -m gpu on an MI355X).  Helpers come from the earlier GPU test modules."""
import numpy as np
import pytest

from oracle import w2l_oracle as o
from test_gpu_parity import _report, rel_l2, weights64
from test_gpu_round4 import _wave_case

pytestmark = pytest.mark.gpu


def test_raw_wave_elu_shorter_batch_after_a_longer_one():
    """with ELU the input-gradient launch of striding_conv leaves non-zero rows of gx0 just past the valid
    frames (elu'(0) = 1), and wave_conv's weight gradient contracts whole 64-row chunks of the gathered sample windows against
    them -- a shorter batch after a longer one in the same buffer set must not see the longer batch's windows
    (_Buffers.set_length clears `frames` beyond the new length).  Front-layer gradients of the SECOND, shorter batch against
    the float64 oracle, and against a fresh engine that never saw the longer batch."""
    import torch
    from speechless_amd.engine import Engine
    case = _wave_case(activation="elu")

    def run(eng, t_audio):
        x = case["x"][:, :t_audio]
        pred_len = [min(n, t_audio // 320) for n in case["prediction_lengths"]]
        eng.load_input(x)
        eng.set_labels(case["labels"], np.array(case["label_lengths"]), np.array(pred_len))
        eng.forward(training=True)
        losses = eng.ctc().cpu().numpy()
        eng.backward()
        torch.cuda.synchronize()
        return x, pred_len, losses, eng.get_gradients()

    eng = Engine(case["specs"], 29, dtype="f32")
    eng.set_weights(case["weights"])
    run(eng, 24055)
    x, pred_len, losses, grads = run(eng, 20007)       # same buffer set (same padded length), 26 frames shorter
    fresh = Engine(case["specs"], 29, dtype="f32")
    fresh.set_weights(case["weights"])
    _, _, losses_fresh, grads_fresh = run(fresh, 20007)
    ref = o.loss_and_gradients(case["ospecs"], weights64(case), x.astype(np.float64), case["labels"], pred_len,
                               case["label_lengths"])
    np.testing.assert_allclose(losses, ref["losses"], rtol=2e-5)
    errs = [max(rel_l2(dw, rw), rel_l2(db, rb)) for (dw, db), (rw, rb) in zip(grads, ref["grads"])]
    _report("raw_wave_elu_second_batch_gradient_errors", errs)
    assert max(errs) < 2e-4, errs                       # ELU: no ReLU decisions to flip, every layer tight
    for (dw, db), (fw, fb) in zip(grads, grads_fresh):
        assert np.array_equal(dw, fw) and np.array_equal(db, fb)
    assert np.array_equal(losses, losses_fresh)



# ------------------------------------------------------------------------------------------ raw-wave input on bf16x3
def _wave_engine(case, dtype):
    from speechless_amd.engine import Engine
    eng = Engine(case["specs"], 29, dtype=dtype)
    eng.set_weights(case["weights"])
    return eng


def _wave_step(eng, case, t_audio):
    import torch
    x = case["x"][:, :t_audio]
    pred_len = [min(n, t_audio // 320) for n in case["prediction_lengths"]]
    eng.load_input(x)
    eng.set_labels(case["labels"], np.array(case["label_lengths"]), np.array(pred_len))
    eng.forward(training=True)
    losses = eng.ctc().cpu().numpy().copy()
    eng.backward()
    torch.cuda.synchronize()
    return x, pred_len, losses, eng.get_gradients()


def test_raw_wave_bf16x3_elu_against_the_float64_oracle():
    """bf16x3 with an ELU wave_conv (round 5): the front layer's pre-activations go through the fp32 staging buffer and
    sl_split3 (mode elu) into the pair-view input, the input gradient of striding_conv through sl_split3 (mode elu mask).  ELU has
    no decisions to flip, so every one of the twelve gradients is tight -- on a long batch and on a shorter one after it."""
    case = _wave_case(activation="elu")
    eng = _wave_engine(case, "bf16x3")
    for t_audio in (24055, 20007):
        x, pred_len, losses, grads = _wave_step(eng, case, t_audio)
        ref = o.loss_and_gradients(case["ospecs"], weights64(case), x.astype(np.float64), case["labels"], pred_len,
                                   case["label_lengths"])
        np.testing.assert_allclose(losses, ref["losses"], rtol=2e-5)
        errs = [max(rel_l2(dw, rw), rel_l2(db, rb)) for (dw, db), (rw, rb) in zip(grads, ref["grads"])]
        _report("raw_wave_bf16x3_elu_gradient_errors_t{}".format(t_audio), errs)
        assert max(errs) < 2e-4, errs


@pytest.mark.parametrize("activation", ["relu", "elu"])
def test_raw_wave_bf16x3_dropout_draws_the_masks_of_the_f32_path(activation):
    """Dropout in front of wave_conv and of the stack's layers (net.py:301-303) on the bf16x3 raw-wave path: the samples are
    dropped in fp32 before the windows are gathered and every plane tensor by sl_split3_dropout with the (seed, element) keep
    decisions of the single-plane kernels -- so the exact-fp32 path with the same seed sees the same masks and the two paths
    agree as they do without dropout (ELU: every gradient tight; ReLU: flip-aware, the top tight)."""
    case = _wave_case(activation=activation)
    out = {}
    for dtype in ("f32", "bf16x3"):
        eng = _wave_engine(case, dtype)
        eng.dropout_rate, eng.dropout_seed = 0.2, 11
        out[dtype] = _wave_step(eng, case, 24055)
        # a second step (new masks) runs too, and evaluation afterwards ignores the rate
        _wave_step(eng, case, 20007)
        probs = eng.forward(case["x"][:, :20007]).cpu().numpy().copy()
        eng.dropout_rate = None
        assert np.array_equal(eng.forward(case["x"][:, :20007]).cpu().numpy(), probs)
    (_, _, l32, g32), (_, _, lx3, gx3) = out["f32"], out["bf16x3"]
    np.testing.assert_allclose(lx3, l32, rtol=2e-5)
    errs = [max(rel_l2(a, c), rel_l2(ab, cb)) for (a, ab), (c, cb) in zip(gx3, g32)]
    _report("raw_wave_bf16x3_dropout_{}_vs_f32".format(activation), errs)
    assert np.isfinite(errs).all() and errs[-1] < 5e-4 and max(errs) < (2e-4 if activation == "elu" else 3e-2), errs


def test_raw_wave_bf16x3_launch_lists_and_rccl_single_rank():
    """Recorded launch lists over batches of two lengths in one buffer set, and the bucketed exchange through RCCL on one rank
    (the front layer's bucket announced last): both bit-identical to the plain eager step on the bf16x3 raw-wave path."""
    import os
    import torch
    import torch.distributed as dist
    from speechless_amd.parallel import GradBucketReducer
    case = _wave_case()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29537")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        finals = []
        for lists, use_reducer in ((True, False), (False, False), (True, True)):
            eng = _wave_engine(case, "bf16x3")
            eng.use_launch_lists = lists
            announced = []
            reducer = None
            if use_reducer:
                reducer = GradBucketReducer(eng.grads, eng.bucket_ranges(), force=True)
                inner = reducer.reduce_bucket
                reducer.reduce_bucket = lambda b, inner=inner: (announced.append(b), inner(b))[1]
            losses = []
            for step in range(4):
                t_audio = 24055 if step % 2 == 0 else 20007
                pred_len = [min(n, t_audio // 320) for n in case["prediction_lengths"]]
                loss = eng.train_step(case["x"][:, :t_audio], case["labels"], np.array(case["label_lengths"]),
                                      np.array(pred_len), reducer)
                losses.append(loss.cpu().numpy().copy())
            torch.cuda.synchronize()
            if use_reducer:
                nb = len(eng.bucket_plan())
                assert eng.bucket_plan()[-1][0] == [eng.front_plan.index] and announced == list(range(nb)) * 4, announced
            finals.append((np.stack(losses), eng.params.clone()))
    finally:
        if created:
            dist.destroy_process_group()
    for other in finals[1:]:
        assert np.array_equal(finals[0][0], other[0]) and torch.equal(finals[0][1], other[1])
    assert np.isfinite(finals[0][0]).all() and finals[0][0][2].mean() < finals[0][0][0].mean()


def test_random_shapes_through_the_raw_wave_topology():
    """tools/fuzz_shapes.py --wave --x3, ten cases: random batch sizes, sample counts (mostly no multiple of the stride) and label
    lengths through the whole optimisation step of the 12-layer raw-wave net on the bf16, fp32 and bf16x3 paths -- finite,
    deterministic, bf16 loss within 2e-3 and bf16x3 loss within 2e-6 of fp32 (round 5: 60 cases passed, worst 3.7e-6 / < 2e-6)."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    res = subprocess.run([sys.executable, str(root / "tools" / "fuzz_shapes.py"), "--wave", "--x3", "--cases", "10", "--seed", "31",
                          "--max-frames", "400", "--max-batch", "4"], capture_output=True, text=True, cwd=str(root), timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert "all 10 cases passed" in res.stdout


def test_wav2letter_api_with_raw_wave_input_on_the_bf16x3_path(tmp_path):
    """Wav2Letter(use_raw_wave_input=True, compute_dtype="bf16x3"): the reference's constructor surface on the fast parity path --
    training steps lower the loss, predict() returns text, an HDF5 checkpoint with the `wave_conv` layer round-trips into the
    fp32 path and both paths then give the same greedy transcription and the same loss to 1e-5."""
    from speechless_amd import Wav2Letter, english_frequent_characters
    from speechless_amd.net import Adam, LabeledSpectrogram
    rng = np.random.RandomState(4)
    words = ["she", "was", "abc", "a", "zoo"]
    batch = [LabeledSpectrogram(id="u{}".format(i), label=" ".join(rng.choice(words, size=rng.randint(1, 3))),
                                spectrogram=0.1 * rng.randn(int(rng.randint(9000, 12000)), 1)) for i in range(4)]
    sizes = dict(out_filter_count=256)
    net = Wav2Letter(1, english_frequent_characters, use_raw_wave_input=True, optimizer=Adam(1e-3), seed=5,
                     compute_dtype="bf16x3", layer_sizes=sizes)
    assert net.input_to_prediction_length_ratio == 320 and net.engine.dtype == "bf16x3"
    before = net.test_and_predict_batch(batch).average_loss
    for _ in range(10):
        net.train_on_batch(batch)
    after = net.test_and_predict_batch(batch)
    assert after.average_loss < before and isinstance(net.predict(batch[0]), str)
    net.predictive_net.save_weights(tmp_path / "w.h5")
    other = Wav2Letter(1, english_frequent_characters, use_raw_wave_input=True, seed=9, compute_dtype="f32", layer_sizes=sizes)
    other.predictive_net.load_weights(str(tmp_path / "w.h5"))
    same = other.test_and_predict_batch(batch)
    assert [r.predicted for r in same.results] == [r.predicted for r in after.results]
    np.testing.assert_allclose([r.loss for r in same.results], [r.loss for r in after.results], rtol=1e-5)
