#!/usr/bin/env python
"""Times sl_conv1d_wgrad_grouped (the 7 inner layers in one launch) over tile configurations / batch splits."""
import ctypes, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from speechless_amd.engine import Engine, wav2letter_layer_specs

def cfg_word(wm, wn, stages, splits): return wm | (wn << 4) | (stages << 8) | (splits << 12)
specs = wav2letter_layer_specs(128, 29)
eng = Engine(specs, 29, dtype="bf16")
rng = np.random.RandomState(0)
eng.set_weights([(rng.uniform(-.05, .05, size=(s.kernel_size, s.cin, s.cout)).astype(np.float32),
                  np.zeros(s.cout, np.float32)) for s in specs])
B, T = 32, 1000
x = rng.randn(B, T, 128).astype(np.float32)
lab_len = rng.randint(20, 201, size=B); labels = np.zeros((B, 200), np.int32)
eng.load_input(x); eng.set_labels(labels, lab_len, np.full(B, T // 2)); eng.forward(); eng.ctc(); eng.backward()
torch.cuda.synchronize()
buf = eng.cur; lo, hi = eng.runs[0]; plo = eng.plans[lo]
dw, _ = eng.layer_param_views(eng.grads, plo)
ws = torch.empty((1 << 30,), dtype=torch.uint8, device=eng.device)
st = torch.cuda.current_stream().cuda_stream
stride = buf.batch * buf.rows * plo.cin_pad
def run(cfg):
    eng.lib.call("sl_conv1d_wgrad_grouped", buf.y[lo - 1].data_ptr(), buf.g[lo].data_ptr(), dw.data_ptr(),
                 ctypes.byref(buf.wgrad_geom[lo]), hi - lo + 1, stride, stride, plo.w_numel + plo.cout_pad, cfg,
                 ws.data_ptr(), ws.numel(), st)
run(0); ref = eng.grads.clone()
flops = 2.0 * B * buf.t_out * 7 * 250 * 250 * (hi - lo + 1)
res = []
for (wm, wn, stg) in [(2, 2, 2), (2, 2, 3), (4, 2, 2), (2, 4, 2), (4, 4, 2), (4, 4, 10)]:
    for sp in (1, 2, 3, 4, 5, 6, 8, 16):
        cfg = cfg_word(wm, wn, stg, sp)
        try:
            run(cfg); torch.cuda.synchronize()
        except Exception as e:
            print(wm, wn, stg, sp, "ERR", str(e)[:80]); continue
        err = float((eng.grads - ref).abs().max() / ref.abs().max())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run(cfg)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        res.append((ms, wm, wn, stg, sp, err))
for r in sorted(res)[:12]: print("%.4f ms  %4.0f TFLOP/s  cfg(wm,wn,stages,splits)=%s err=%.1e" % (r[0], flops / r[0] / 1e9, r[1:5], r[5]))
