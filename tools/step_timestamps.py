#!/usr/bin/env python
"""s_memtime stamps of one steady-state step of the interleaved tap-major NT kernel (wave 0 of work-group 0): where a
step's cycles go -- phase A + its fragment-read wait, the request (vmcnt) wait, the barrier, phase B + the wait at the
top of the next step.  Needs a probe build:  SL_EXTRA_FLAGS=-DSL_PROBE_TIMES python -m speechless_amd.build --force
(the stamps land in the split-K workspace, which the probed launches do not use)."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, "/root/repo" if False else ".")
from speechless_amd import _lib
from speechless_amd.engine import Engine, wav2letter_layer_specs
specs = wav2letter_layer_specs(128, 29)
eng = Engine(specs, 29, dtype="bf16")
rng = np.random.RandomState(0)
eng.set_weights([(rng.uniform(-0.05, 0.05, size=(s.kernel_size, s.cin, s.cout)).astype(np.float32), np.zeros(s.cout, np.float32)) for s in specs])
eng.load_input(rng.randn(32, 1000, 128).astype(np.float32))
eng.forward()
torch.cuda.synchronize()
buf = eng.cur
st = torch.cuda.current_stream().cuda_stream
ws = torch.zeros((1 << 20,), dtype=torch.uint8, device=eng.device)
for name, i in (("inner_conv_3", 3), ("striding_conv", 0), ("big_conv_2", 9)):
    p = eng.plans[i]
    xin = buf.x0 if i == 0 else buf.y[i - 1]
    _, bias = eng.layer_param_views(eng.params, p)
    ws.zero_()
    for _ in range(3):
        eng.lib.call("sl_conv1d_nt", xin.data_ptr(), eng.w_fwd[i].data_ptr(), bias.data_ptr(), None, buf.y[i].data_ptr(),
                     ctypes.byref(buf.fwd_geom[i]), _lib.EPI_BIAS_RELU, eng.dtype_code, 0, 0, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    t = ws[:48].cpu().numpy().view(np.uint64)
    t1, t3, t4, t5, n1, n = [int(v) for v in t[:6]]
    print(name, "steps", n, "| phase A + read wait", t3 - t1, "| request wait", t4 - t3, "| barrier", t5 - t4, "| phase B + top wait", n1 - t5, "| step", n1 - t1)
