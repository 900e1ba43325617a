#!/usr/bin/env python
"""Utterances/s of the resident training step (BASELINE config-3 shapes) over the per-GPU batch size: shows how far the
tile tables, which were measured at B = 32, carry to other batch sizes.    python tools/batch_sweep.py"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    from speechless_amd.net import Wav2Letter
    specs = wav2letter_layer_specs(128, 29)
    flops_per_utt = 72.4335e9
    for batch in (4, 8, 16, 32, 48, 64, 128):
        eng = Engine(specs, 29, dtype="bf16")
        eng.set_weights(Wav2Letter._glorot_uniform(specs, 2))
        rng = np.random.RandomState(0)
        x = rng.randn(batch, 1000, 128).astype(np.float32)
        lab_len = rng.randint(20, 201, size=batch)
        labels = -np.ones((batch, 200), dtype=np.int32)
        for i, n in enumerate(lab_len):
            labels[i, :n] = rng.randint(0, 28, size=n)
        eng.load_input(x)
        eng.set_labels(labels, lab_len, np.full(batch, 500))
        for _ in range(5):
            eng.train_step_resident()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        steps = 30
        for _ in range(steps):
            eng.train_step_resident()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / steps
        print("B = {:3d}: {:7.3f} ms/step  {:8.0f} utt/s  {:5.1f} % of the bf16 peak".format(
            batch, ms, batch / ms * 1e3, batch / ms * 1e3 * flops_per_utt / 2.5e15 * 100))
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
