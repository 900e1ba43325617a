import sys, time
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
def timed(label, steps=30):
    for _ in range(5): eng.train_step_resident()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): eng.train_step_resident()
    b.record(); torch.cuda.synchronize()
    print("%-40s %.4f ms/step" % (label, a.elapsed_time(b) / steps))
eng.early_adam = False; timed("adam after backward")
eng.early_adam = True; timed("early adam (side stream)")
real = eng._adam_layers
eng._adam_layers = lambda layers, st: None
timed("no adam at all")
eng._adam_layers = real
real_launch = eng._launch
def no_bias(tag, name, *args):
    if name == "sl_bias_grad": return
    real_launch(tag, name, *args)
eng._launch = no_bias
timed("early adam, no bias grads")
eng._adam_layers = lambda layers, st: None
timed("no adam, no bias grads")
def only_conv(tag, name, *args):
    if name in ("sl_bias_grad", "sl_ctc_loss_grad", "sl_softmax_logq"): return
    real_launch(tag, name, *args)
eng._launch = only_conv
timed("convs only")
