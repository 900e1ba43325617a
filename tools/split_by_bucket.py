#!/usr/bin/env python
"""Config 5, bucket by bucket: the training step with Engine.split_top on / off (which batch geometries gain from running the
CTC of one half-batch under the top layers of the other, and which lose because half-batch launches leave the chip empty)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    import bench

    class A:
        profile_steps = 1
    b = bench.Bench(5, A(), 1, 0, "cuda:0")
    eng = b.eng
    # default: equal halves on every bucket (the mechanism, whatever the engine's rule says); --rule: the engine's own choice
    eng.split_min_tiles = None if "--rule" in sys.argv else 0
    print("{:>6} {:>6} {:>9} {:>9} {:>8} {:>6}".format("T'max", "tiles", "whole_ms", "split_ms", "diff", "part"))
    for (x_dev, lab, ll, pl, tl) in b.resident:
        res = {}
        for split in (False, True, False, True):
            eng.split_top = split
            for _ in range(3):
                eng.load_input(x_dev)
                eng.set_labels(lab, ll, pl)
                eng.train_step_resident()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                eng.load_input(x_dev)
                eng.set_labels(lab, ll, pl)
                eng.train_step_resident()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(split, []).append(e0.elapsed_time(e1) / 10)
        t_out = eng.cur.t_out
        tiles = eng.cur.batch * (-(-t_out // 256)) * 8
        w, s = min(res[False]), min(res[True])
        eng.split_top = True
        print("{:6d} {:6d} {:9.4f} {:9.4f} {:+8.4f} {:6d}".format(t_out, tiles, w, s, s - w, eng.split_top_plan(eng.cur)))


if __name__ == "__main__":
    main()
