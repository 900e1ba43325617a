import sys; sys.path.insert(0,'/root/repo')
import torch, bench, numpy as np
from speechless_amd.engine import Engine, wav2letter_layer_specs
from speechless_amd.net import Wav2Letter
specs = wav2letter_layer_specs(bench.MEL, bench.K_CLASSES)
eng = Engine(specs, bench.K_CLASSES, dtype="bf16x3"); eng.set_weights(Wav2Letter._glorot_uniform(specs, 2))
x, labels, lab_len, pred_len = bench.synthetic_batch(0, 32)
eng.load_input(torch.from_numpy(x).cuda()); eng.set_labels(labels, lab_len, pred_len)
for _ in range(3): eng.train_step_resident()
eng.timeline = []
for _ in range(3): eng.train_step_resident()
torch.cuda.synchronize()
agg = {}
for tag, a, b in eng.timeline:
    k = tag.split(":")[0] if not tag.startswith("split:dgrad") else "split_dgrad"
    agg[k] = agg.get(k, 0) + a.elapsed_time(b) / 3
print({k: round(v, 3) for k, v in sorted(agg.items(), key=lambda e: -e[1])}, round(sum(agg.values()), 3))
