import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from speechless_amd.engine import Engine, wav2letter_layer_specs
specs = wav2letter_layer_specs(128, 29)
eng = Engine(specs, 29, dtype="bf16")
rng = np.random.RandomState(0)
ws = []
for s in specs:
    lim = np.sqrt(6.0 / (s.kernel_size * (s.cin + s.cout)))
    ws.append((rng.uniform(-lim, lim, size=(s.kernel_size, s.cin, s.cout)).astype(np.float32), np.zeros(s.cout, np.float32)))
eng.set_weights(ws)
B = 32
x = rng.randn(B, 1000, 128).astype(np.float32)
lab_len = rng.randint(20, 201, size=B)
labels = -np.ones((B, 200), dtype=np.int32)
for i, n in enumerate(lab_len):
    labels[i, :n] = rng.randint(0, 28, size=n)
eng.load_input(x)
eng.set_labels(labels, lab_len, np.full(B, 500))
import statistics
def timed(steps=40):
    for _ in range(3): eng.train_step_resident()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): eng.train_step_resident()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
real_adam = eng._adam_layers
real_launch = eng._launch
def no_bias(tag, name, *args):
    if name == "sl_bias_grad": return
    real_launch(tag, name, *args)
def only_conv(tag, name, *args):
    if name in ("sl_bias_grad", "sl_ctc_loss_grad", "sl_softmax_logq"): return
    real_launch(tag, name, *args)
variants = {
    "A default (Adam after backward, bias grads on side stream)": dict(early=False, adam=True, launch=real_launch, ow=False),
    "C no adam": dict(early=False, adam=False, launch=real_launch, ow=False),
    "D no bias grads": dict(early=False, adam=True, launch=no_bias, ow=False),
    "F convs only": dict(early=False, adam=False, launch=only_conv, ow=False),
}
res = {k: [] for k in variants}
for rep in range(4):
    for k, v in variants.items():
        eng._adam_layers = real_adam if v["adam"] else (lambda layers, st: None)
        eng._launch = v["launch"]
        res[k].append(timed())
for k, v in res.items():
    print("%-62s median %.4f  min %.4f  all %s" % (k, statistics.median(v), min(v), [round(x, 3) for x in v]))
