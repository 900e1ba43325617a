import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from speechless_amd.engine import Engine, wav2letter_layer_specs
from speechless_amd.net import Wav2Letter
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
specs = wav2letter_layer_specs(128, 29)
eng = Engine(specs, 29, dtype=dtype)
eng.set_weights(Wav2Letter._glorot_uniform(specs, 2))
rng = np.random.RandomState(0)
b = 32
x = rng.randn(b, 1000, 128).astype(np.float32)
lab_len = rng.randint(20, 201, size=b)
labels = -np.ones((b, 200), dtype=np.int32)
for i, n in enumerate(lab_len):
    labels[i, :n] = rng.randint(0, 28, size=n)
eng.load_input(x); eng.set_labels(labels, lab_len, np.full(b, 500))
for _ in range(5): eng.train_step_resident()
torch.cuda.synchronize()
eng.timeline = []
n = 5
for _ in range(n): eng.train_step_resident()
torch.cuda.synchronize()
per = {}
for tag, a, c in eng.timeline: per.setdefault(tag, []).append(a.elapsed_time(c))
eng.timeline = None
tot = 0
for tag, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    ms = sum(v) / n; tot += ms
    print("%-45s %.4f" % (tag, ms))
print("sum", tot)
