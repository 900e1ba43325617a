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

