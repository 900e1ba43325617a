#!/usr/bin/env python
"""ms per resident config-3 training step (median of --reps runs of --steps steps): for A/B runs of two LIBRARY builds on one
box (SL_LIB_PATH=tools/_probe/lib_X.so python tools/step_ms.py), alternated by tools/lib_ab.sh."""
import argparse
import statistics
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    import torch
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    from speechless_amd.net import Wav2Letter
    specs = wav2letter_layer_specs(128, 29)
    eng = Engine(specs, 29, dtype=args.dtype)
    eng.set_weights(Wav2Letter._glorot_uniform(specs, 2))
    rng = np.random.RandomState(0)
    b = 32
    x = rng.randn(b, 1000, 128).astype(np.float32)
    lab_len = rng.randint(20, 201, size=b)
    labels = -np.ones((b, 200), dtype=np.int32)
    for i, n in enumerate(lab_len):
        labels[i, :n] = rng.randint(0, 28, size=n)
    eng.load_input(x)
    eng.set_labels(labels, lab_len, np.full(b, 500))
    for _ in range(10):
        eng.train_step_resident()
    torch.cuda.synchronize()
    out = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            eng.train_step_resident()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / args.steps)
    print("{:.4f} ms median  (min {:.4f})  {}".format(statistics.median(out), min(out), [round(v, 4) for v in out]))


if __name__ == "__main__":
    main()
