#!/usr/bin/env python
"""With a -DSL_CHAIN_PROBE_TIMES build (SL_LIB_PATH): s_memtime stamps of every wave of the fused inner-layer launch
(forward) at BASELINE config-3 shape -> where the time between the layers goes."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    import bench
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    from speechless_amd.net import Wav2Letter
    from speechless_amd._lib import lib
    specs = wav2letter_layer_specs(bench.MEL, bench.K_CLASSES)
    x, labels, lab_len, pred_len = bench.synthetic_batch(0, bench.BATCH_PER_GPU)
    eng = Engine(specs, bench.K_CLASSES, dtype="bf16")
    eng.use_chain = True
    eng.use_launch_lists = False
    eng.set_weights(Wav2Letter._glorot_uniform(specs, 2))
    eng.load_input(torch.from_numpy(x).cuda())
    eng.set_labels(labels, lab_len, pred_len)
    direction = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    for _ in range(3):
        eng.forward(training=True)
        if direction != "fwd":
            eng.ctc()
            eng.backward()
    torch.cuda.synchronize()
    raw = ctypes.CDLL(os.environ["SL_LIB_PATH"])
    out = np.zeros(256 * 8 * 40, dtype=np.uint64)
    rc = raw.sl_chain_probe_read(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.nbytes))
    assert rc == 0, rc
    t = out.reshape(256, 8, 40).astype(np.int64)
    base = t[:, :, 0].min(axis=1, keepdims=True)[:, :, None]
    rel = t - base
    print("direction", direction, "; cycles of the s_memtime counter, mean over 256 work-groups x 8 waves (backward: 'barrier-2 wait' includes the ReLU-mask pass)")
    print("begin -> input rows in LDS: {:.0f}".format((rel[:, :, 1]).mean()))
    prev = rel[:, :, 1]
    tot = dict(steps=0., wait1=0., epi=0., wait2=0.)
    for l in range(7):
        s0, s1, s2, s3 = (rel[:, :, 2 + 4 * l + i] for i in range(4))
        print("layer {}: steps {:7.0f}  (per step {:5.0f}; slowest-fastest wave {:5.0f})  barrier-1 wait {:5.0f}  epilogue {:5.0f}  "
              "barrier-2 wait {:5.0f}".format(l, (s0 - prev).mean(), (s0 - prev).mean() / 28,
                                              (s0.max(axis=1) - s0.min(axis=1)).mean(), (s1 - s0).mean(), (s2 - s1).mean(),
                                              (s3 - s2).mean()))
        tot["steps"] += (s0 - prev).mean(); tot["wait1"] += (s1 - s0).mean(); tot["epi"] += (s2 - s1).mean()
        tot["wait2"] += (s3 - s2).mean()
        prev = s3
    end = rel[:, :, 38]
    print("total {:.0f} cycles; steps {:.0f}, barrier-1 {:.0f}, epilogue {:.0f}, barrier-2 {:.0f}".format(
        end.mean(), tot["steps"], tot["wait1"], tot["epi"], tot["wait2"]))
    print("whole launch (first begin -> last end): {:.0f} cycles".format(float(t[:, :, 38].max() - t[:, :, 0].min())))


if __name__ == "__main__":
    main()
