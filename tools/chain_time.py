#!/usr/bin/env python
"""Times the run of inner_conv_1..7 at BASELINE config-3 shape: sl_conv1d_chain (one launch) against the seven
sl_conv1d_nt launches, forward and input-gradient direction, back to back.   (SL_LIB_PATH selects a probe build)"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    import bench
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    from speechless_amd.net import Wav2Letter
    specs = wav2letter_layer_specs(bench.MEL, bench.K_CLASSES)
    x, labels, lab_len, pred_len = bench.synthetic_batch(0, bench.BATCH_PER_GPU)
    for chain in (False, True):
        eng = Engine(specs, bench.K_CLASSES, dtype="bf16")
        eng.use_chain = chain
        eng.use_launch_lists = False
        eng.set_weights(Wav2Letter._glorot_uniform(specs, 2))
        eng.load_input(torch.from_numpy(x).cuda())
        eng.set_labels(labels, lab_len, pred_len)
        for _ in range(3):
            eng.forward(training=True)
            eng.ctc()
            eng.backward()
        torch.cuda.synchronize()
        eng.timeline = []
        for _ in range(10):
            eng.forward(training=True)
            eng.ctc()
            eng.backward()
        torch.cuda.synchronize()
        per = {}
        for tag, a, b in eng.timeline:
            if "inner" in tag and (tag.startswith("fwd") or tag.startswith("dgrad")):
                per.setdefault(tag.split(":")[0], []).append(a.elapsed_time(b))
        eng.timeline = None
        print("chain" if chain else "single launches", {k: round(float(np.sum(v)) / 10 * 1e3, 1) for k, v in per.items()},
              "us per step (events around each launch)")


if __name__ == "__main__":
    main()
