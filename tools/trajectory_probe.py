#!/usr/bin/env python
"""Loss trajectories of the bf16 path, the fp32 path and the fp32 path from minutely perturbed initial weights on one
synthetic batch (is the bf16 / fp32 gap bias, or the chaos of the optimisation itself?).  Writes gpurun_out/trajectory.json.
    python tools/trajectory_probe.py [--batch 32] [--frames 200] [--steps 300] [--lr 1e-4]"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--perturb", type=float, default=1e-6)
    args = ap.parse_args()
    import torch
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    from speechless_amd.net import Wav2Letter
    specs = wav2letter_layer_specs(128, 29)
    weights = Wav2Letter._glorot_uniform(specs, 2)
    rng = np.random.RandomState(78)
    x = np.random.RandomState(77).randn(args.batch, args.frames, 128).astype(np.float32)
    t_out = args.frames // 2
    lab_len = rng.randint(5, max(6, t_out // 3), size=args.batch).astype(np.int32)
    labels = -np.ones((args.batch, int(lab_len.max())), dtype=np.int32)
    for i, n in enumerate(lab_len):
        labels[i, :n] = rng.randint(0, 28, size=n)
    pred_len = np.full((args.batch,), t_out, dtype=np.int32)
    prng = np.random.RandomState(5)
    perturbed = [(w * (1 + args.perturb * prng.randn(*w.shape)).astype(np.float32), b) for w, b in weights]
    curves = {}
    for name, dtype, w in (("f32", "f32", weights), ("f32_perturbed", "f32", perturbed), ("bf16", "bf16", weights),
                           ("bf16_perturbed", "bf16", perturbed)):
        eng = Engine(specs, 29, dtype=dtype, lr=args.lr)
        eng.set_weights(w)
        eng.load_input(x)
        eng.set_labels(labels, lab_len, pred_len)
        means = [eng.train_step_resident().mean() for _ in range(args.steps)]
        torch.cuda.synchronize()
        curves[name] = [float(m.item()) for m in means]
        print(name, " ".join("{:.2f}".format(v) for v in curves[name][::max(1, args.steps // 20)]), flush=True)
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "trajectory.json").write_text(json.dumps({"args": vars(args), "curves": curves}))


if __name__ == "__main__":
    main()
