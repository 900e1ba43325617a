"""Do results depend on what the allocator hands out?  Same run on fresh memory, then on memory full of NaN patterns."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch
from speechless_amd.engine import Engine, wav2letter_layer_specs
from speechless_amd.net import Wav2Letter

def run(dtype, b, t, steps, bins=128):
    specs = wav2letter_layer_specs(bins, 29)
    weights = Wav2Letter._glorot_uniform(specs, 2)
    rng = np.random.RandomState(78)
    x = np.random.RandomState(77).randn(b, t, bins).astype(np.float32)
    lab_len = rng.randint(5, 31, size=b).astype(np.int32)
    labels = -np.ones((b, int(lab_len.max())), dtype=np.int32)
    for i, n in enumerate(lab_len):
        labels[i, :n] = rng.randint(0, 28, size=n)
    eng = Engine(specs, 29, dtype=dtype)
    eng.set_weights(weights)
    eng.load_input(x)
    eng.set_labels(labels, lab_len, np.full((b,), t // 2, dtype=np.int32))
    out = [eng.train_step_resident().clone() for _ in range(steps)]
    torch.cuda.synchronize()
    g = eng.grads.clone()
    return torch.stack(out).cpu().numpy(), g.cpu().numpy()

def poison(gb):
    blocks = []
    for size in (1 << 30, 256 << 20, 64 << 20, 16 << 20, 4 << 20, 1 << 20, 256 << 10, 64 << 10, 4096, 512):
        n = max(4, min(64, int(gb * (1 << 30) / 10 / size)))
        for _ in range(n):
            blocks.append(torch.full((size // 4,), float("nan"), dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    del blocks

for dtype, b, t, steps in (("bf16", 32, 200, 12), ("bf16", 4, 96, 6), ("bf16", 32, 1000, 4), ("f32", 8, 200, 4), ("bf16x3", 8, 200, 4)):
    torch.cuda.empty_cache()
    la, ga = run(dtype, b, t, steps)
    poison(20)
    lb, gb_ = run(dtype, b, t, steps)
    poison(20)
    lc, gc = run(dtype, b, t, steps)
    same1 = np.array_equal(la, lb, equal_nan=True) and np.array_equal(ga, gb_, equal_nan=True)
    same2 = np.array_equal(lb, lc, equal_nan=True) and np.array_equal(gb_, gc, equal_nan=True)
    first = [int(i) for i in np.nonzero((la != lb).any(axis=1))[0][:3]]
    print(dtype, b, t, "fresh == poisoned:", same1, "poisoned == poisoned:", same2, "first differing steps", first,
          "nan in losses", bool(np.isnan(lb).any()), "max |dloss|", float(np.nanmax(np.abs(la - lb))))
