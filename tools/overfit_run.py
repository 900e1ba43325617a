#!/usr/bin/env python
"""3000 resident training steps on ONE fixed synthetic batch (config-3 shape, bf16 path, every default of the round: fused
inner-layer launches, ones-channel bias gradients, wave CTC lattice): the loss must fall from ~1300 to ~1e-3 through the
blank-collapse phase without a non-finite value, a slow (repaired) step or a padded weight leaving zero.
-> profiles/r03_overfit_run.txt (r02k: the same run with round 2's lattice, cumulative averages)"""
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
torch.cuda.synchronize()
t0, last = time.perf_counter(), -1
for step in range(3001):
    loss = eng.train_step_resident()
    if step % 250 == 0:
        l = loss.cpu().numpy()  # (synchronises)
        dt = (time.perf_counter() - t0) / (step - last) * 1e3  # over the steps since the previous line
        probs = eng.cur.probs.cpu().numpy()
        blank = probs[..., -1].mean()
        print("step %4d  mean loss %9.3f  finite %s  mean p(blank) %.4f  %.3f ms/step over the last %d" % (
            step, l.mean(), np.isfinite(l).all(), blank, dt, step - last), flush=True)
        torch.cuda.synchronize()
        t0, last = time.perf_counter(), step
w = eng.get_weights()
print("weights finite:", all(np.isfinite(a).all() and np.isfinite(b).all() for a, b in w))
for p in eng.plans:
    wv, bv = eng.layer_param_views(eng.params, p)
    assert not wv[:, p.spec.cin:, :].any() and not wv[:, :, p.spec.cout:].any()
print("padded weights still zero")
