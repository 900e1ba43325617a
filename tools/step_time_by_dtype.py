#!/usr/bin/env python
"""Resident training step (config 3 batch: 32 x 1000 frames, labels U{20..200}) and forward + decode (config 2) per storage
scheme: bf16 (benchmarked), bf16x3 / f16x3 (hi + lo bf16 / fp16 planes, three MFMA terms: the fast parity paths), f32 (exact-fp32
MFMA).

    python tools/step_time_by_dtype.py [--steps 10]
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    import bench
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    from speechless_amd.net import Wav2Letter
    specs = wav2letter_layer_specs(bench.MEL, bench.K_CLASSES)
    weights = Wav2Letter._glorot_uniform(specs, 2)
    x, labels, lab_len, pred_len = bench.synthetic_batch(0, bench.BATCH_PER_GPU)
    for dtype in ("bf16", "bf16x3", "f16x3", "f32"):
        eng = Engine(specs, bench.K_CLASSES, dtype=dtype)
        eng.set_weights(weights)
        eng.load_input(torch.from_numpy(x).cuda())
        eng.set_labels(labels, lab_len, pred_len)

        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / args.steps * 1e3

        def fwd():
            eng.forward()
            eng.greedy_decode()
        f = timed(fwd)
        s = timed(eng.train_step_resident)
        print("{:7s} forward + decode {:7.3f} ms = {:8.0f} utt/s | training step {:7.3f} ms = {:8.0f} utt/s".format(
            dtype, f, 32e3 / f, s, 32e3 / s), flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
