import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from speechless_amd import Wav2Letter, english_frequent_characters
from speechless_amd.spectrogram import LabeledExample
rng = np.random.RandomState(0)
pool = [LabeledExample(lambda a=(0.1 * rng.randn(128000)).astype(np.float32): a, id=str(i), label="ab") for i in range(32)]
net = Wav2Letter(128, english_frequent_characters, seed=0)
ex = net._audio_extractor(pool[0])
flat, offsets, lengths = ex.flatten([e.get_raw_audio() for e in pool])
dev = [torch.from_numpy(a).cuda() for a in (flat, offsets, lengths)]
for _ in range(3):
    ex.batch_device(dev[0], dev[1], dev[2], lengths)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(50):
    ex.batch_device(dev[0], dev[1], dev[2], lengths)
e1.record()
torch.cuda.synchronize()
print("wall %.3f ms, gpu events %.3f ms per batch" % ((time.perf_counter() - t0) / 50 * 1e3, e0.elapsed_time(e1) / 50))
