import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from speechless_amd.spectrogram import SpectrogramExtractor
from speechless_amd import _lib
ext = SpectrogramExtractor()
rng = np.random.RandomState(0)
audios = [(0.1*rng.randn(128000)).astype(np.float32) for _ in range(32)]
flat, off, lens = ext.flatten(audios)
dev = [torch.from_numpy(a).cuda() for a in (flat, off, lens)]
real_call = ext.lib.call
times = {}
def timed_call(name, *args):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); real_call(name, *args); b.record()
    times.setdefault(name, []).append((a, b))
class L:  # proxy
    def __init__(s, lib): s._l = lib
    def call(s, name, *args): timed_call(name, *args)
    def raw(s, name): return s._l.raw(name)
ext.lib = L(ext.lib)
for _ in range(5): ext.batch_device(dev[0], dev[1], dev[2], lens)
torch.cuda.synchronize()
for k, v in times.items():
    print(k, round(float(np.mean([a.elapsed_time(b) for a, b in v[2:]]))*1e3, 1), "us")
