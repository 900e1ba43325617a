#!/usr/bin/env python
"""How idle time converts into step time on a power-limited chip: the resident training step with an idle kernel
(torch.cuda._sleep) of 40 / 85 / 170 us behind it, against the step alone and the idle kernel alone -- the step part stays
what it was, i.e. idle (and low-power) time converts 1 : 1."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from speechless_amd.engine import Engine, wav2letter_layer_specs
from speechless_amd.net import Wav2Letter
specs = wav2letter_layer_specs(bench.MEL, bench.K_CLASSES)
x, labels, lab_len, pred_len = bench.synthetic_batch(0, bench.BATCH_PER_GPU)
eng = Engine(specs, bench.K_CLASSES, dtype="bf16")
eng.set_weights(Wav2Letter._glorot_uniform(specs, 2))
eng.load_input(torch.from_numpy(x).cuda())
eng.set_labels(labels, lab_len, pred_len)
def run(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def sleep_only(c):
    return run(lambda: torch.cuda._sleep(c))
for rep in range(2):
    base = run(eng.train_step_resident)
    print("step alone %.4f ms" % base)
    for c in (100000, 200000, 400000):
        s = sleep_only(c)
        both = run(lambda: (eng.train_step_resident(), torch.cuda._sleep(c)))
        print("  sleep %d cycles: alone %.4f ms; step+sleep %.4f ms; step part = %.4f (delta vs alone %+.4f)" % (c, s, both, both - s, both - s - base))
