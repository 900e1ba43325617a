#!/usr/bin/env python
"""Random shapes through the whole step (forward, CTC, backward, Adam) on the bf16 and the fp32 path of the REAL topology:
batch 1..6, 20..1400 input frames (odd and even, across the 256-frame buffer buckets), ragged input lengths, labels from
empty to the longest the frames allow.  Checks per case: finite losses and gradients, bf16 loss within 2e-3 of fp32, the
same step twice = the same bits, rows outside the utterances untouched.  A geometry bug shows up here as a NaN, a mismatch
or a library error long before it shows up in a benchmark.

    python tools/fuzz_shapes.py [--cases 60] [--seed 0]"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--bins", type=int, default=128, help="input bins (257: the long-form configuration)")
    ap.add_argument("--max-frames", type=int, default=1400)
    ap.add_argument("--max-batch", type=int, default=6)
    ap.add_argument("--x3", action="store_true", help="also the bf16x3 and f16x3 paths (loss within 2e-6 of fp32)")
    ap.add_argument("--split", action="store_true",
                    help="also the bf16 path with Engine.split_top forced on every batch of >= 2 utterances, in a random split "
                         "a + (B - a) (the CTC of one part under the top layers of the other): loss within 2e-5 of the "
                         "whole-batch bf16 step, gradients within 1e-2, deterministic")
    ap.add_argument("--wave", action="store_true",
                    help="the raw-wave topology (use_raw_wave_input: wave_conv over --bins sample channels, 256 output filters in "
                         "the wide layers to keep it quick): --max-frames then counts SAMPLES / 100 (1400 -> up to 140 000)")
    args = ap.parse_args()
    import torch
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    from speechless_amd.net import Wav2Letter
    if args.wave:
        if args.bins > 4:
            args.bins = 1
        specs = wav2letter_layer_specs(args.bins, 29, use_raw_wave_input=True, out_filter_count=256)
    else:
        specs = wav2letter_layer_specs(args.bins, 29)
    weights = Wav2Letter._glorot_uniform(specs, 2)
    engines = {}
    for dtype in ("bf16", "f32") + (("bf16x3",) if args.x3 else ()) + (("f16x3",) if (args.x3 and not args.wave) else ()):
        eng = engines[dtype] = Engine(specs, 29, dtype=dtype)
        eng.max_cached_shapes = 4
    rng = np.random.RandomState(args.seed)
    worst = 0.0
    for case in range(args.cases):
        b = int(rng.randint(1, args.max_batch + 1))
        t = int(rng.choice([rng.randint(20, 80), rng.randint(80, 600), rng.randint(600, max(601, args.max_frames))]))
        t_out = -(-t // 2)
        if args.wave:  # t input frames of the stack <- a sample count that is (mostly) no multiple of the stride
            t = t * 160 - int(rng.randint(0, 160))
            t_out = -(-(-(-t // 160)) // 2)
        x = rng.randn(b, t, args.bins).astype(np.float32)
        t_max_pred = t_out if not args.wave else max(1, t // 320)
        pred_len = np.array([int(rng.randint(max(1, t_max_pred // 3), t_max_pred + 1)) for _ in range(b)], dtype=np.int32)
        lab_len = np.array([int(rng.randint(0, max(1, min(200, p // 2)) + 1)) for p in pred_len], dtype=np.int32)
        labels = -np.ones((b, max(1, int(lab_len.max()))), dtype=np.int32)
        for i, n in enumerate(lab_len):
            labels[i, :n] = rng.randint(0, 28, size=n)
        out = {}
        for dtype, eng in engines.items():
            runs = []
            for rep in range(2):
                eng.set_weights(weights)
                eng.adam_iterations = 0
                eng.adam_m.zero_()
                eng.adam_v.zero_()
                loss = eng.train_step(x, labels, lab_len, pred_len).cpu().numpy().copy()
                torch.cuda.synchronize()
                runs.append((loss, eng.grads.clone(), eng.params.clone()))
            assert np.array_equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]) and \
                torch.equal(runs[0][2], runs[1][2]), ("not deterministic", dtype, b, t)
            loss, grads, params = runs[0]
            assert np.isfinite(loss).all(), (dtype, b, t, loss)
            assert bool(torch.isfinite(grads).all()) and bool(torch.isfinite(params).all()), (dtype, b, t)
            out[dtype] = loss
        if args.split and b >= 2:
            eng = engines["bf16"]
            a = int(rng.randint(1, b))
            whole_grads = None
            runs = []
            for rep in range(3):
                eng.set_weights(weights)
                eng.load_input(x)
                eng.set_labels(labels, lab_len, pred_len)
                if rep == 0:  # the whole-batch sequence
                    eng.forward(training=True)
                    loss = eng.ctc(grad_scale=1.0 / b)
                else:
                    eng.forward(training=True, split_ctc=(1.0 / b, a))
                    loss = eng.cur.loss
                eng.backward()
                torch.cuda.synchronize()
                runs.append((loss.cpu().numpy().copy(), eng.grads.clone()))
            assert np.array_equal(runs[1][0], runs[2][0]) and torch.equal(runs[1][1], runs[2][1]), ("split not deterministic", b, t, a)
            np.testing.assert_allclose(runs[1][0], runs[0][0], rtol=2e-5, atol=1e-4, err_msg=str((b, t, a)))
            err = float(torch.linalg.norm(runs[1][1] - runs[0][1]) / torch.linalg.norm(runs[0][1]))
            assert err < 1e-2, ("split gradients", b, t, a, err)
        rel = float(np.max(np.abs(out["bf16"] - out["f32"]) / np.maximum(np.abs(out["f32"]), 1.0)))
        worst = max(worst, rel)
        assert rel < 2e-3, (b, t, pred_len, lab_len, out)
        if args.x3:
            for x3 in ("bf16x3", "f16x3"):
                if x3 in out:
                    rel3 = float(np.max(np.abs(out[x3] - out["f32"]) / np.maximum(np.abs(out["f32"]), 1.0)))
                    assert rel3 < 2e-6, (x3, b, t, rel3, out)
        print("case %3d  b %d  t %4d  labels %s  loss %s  bf16 vs f32 %.1e" % (
            case, b, t, lab_len.tolist(), np.round(out["f32"], 2).tolist(), rel), flush=True)
    print("all %d cases passed; worst bf16-vs-f32 loss difference %.2e" % (args.cases, worst))


if __name__ == "__main__":
    main()
