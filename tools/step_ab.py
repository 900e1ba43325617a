#!/usr/bin/env python
"""Interleaved same-box A/B of the resident config-3 training step over one Engine attribute.

    python tools/step_ab.py --attr small_bias_pass_on_main [--reps 6] [--steps 40]
    attributes: small_bias_pass_on_main, use_wgrad_multi, fuse_output_backward, use_chain, ones_channel, group_wgrad
Boxes differ by +-4 %, a change of 0.5 % shows only when both settings alternate on the SAME box within one process."""
import argparse
import statistics
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--attr", required=True)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--config", type=int, default=3, choices=(3, 5),
                    help="3: the resident 32 x 1000-frame step; 5: bench.py's long-form bucketed batches (8 x 2000..8000 x 257)")
    ap.add_argument("--force-split", action="store_true", help="split_top on every geometry (not only where the rule says)")
    ap.add_argument("--dtype", default="bf16", help="config 3 only: bf16 | bf16x3 | f32")
    args = ap.parse_args()
    import torch
    if args.config == 5:
        return config5(args)
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
    assert isinstance(getattr(eng, args.attr), bool), args.attr
    if args.force_split:
        eng.split_min_tiles = 0

    def timed():
        for _ in range(4):
            eng.train_step_resident()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            eng.train_step_resident()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.steps

    res = {True: [], False: []}
    for _ in range(args.reps):
        for value in (True, False):
            setattr(eng, args.attr, value)
            for buf in eng._buffers.values():  # (attributes that change the launch sequence: lists and job tables are rebuilt)
                buf.launch_lists, buf.multi_tables = {}, {}
            res[value].append(timed())
    for value in (True, False):
        v = res[value]
        print("{} = {!s:5}  median {:.4f} ms  min {:.4f}  all {}".format(args.attr, value, statistics.median(v), min(v),
                                                                      [round(t, 3) for t in v]))
    print("difference of medians (True - False): {:+.4f} ms".format(statistics.median(res[True]) - statistics.median(res[False])))


def config5(args):
    import statistics as st
    import torch
    import bench

    class A:
        profile_steps = 1
    b = bench.Bench(5, A(), 1, 0, "cuda:0")
    eng = b.eng
    assert isinstance(getattr(eng, args.attr), bool), args.attr
    n = len(b.resident)
    if args.force_split:
        eng.split_min_tiles = 0

    def timed():
        b.cursor = 0
        for _ in range(n):
            b.step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2 * n):
            b.step()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (2 * n)
    res = {True: [], False: []}
    for _ in range(args.reps):
        for value in (True, False):
            setattr(eng, args.attr, value)
            res[value].append(timed())
    for value in (True, False):
        v = res[value]
        print("config 5: {} = {!s:5}  median {:.4f} ms  min {:.4f}  all {}".format(args.attr, value, st.median(v), min(v),
                                                                                [round(t, 3) for t in v]))
    print("config 5: difference of medians (True - False): {:+.4f} ms".format(st.median(res[True]) - st.median(res[False])))


if __name__ == "__main__":
    main()
