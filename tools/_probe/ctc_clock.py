import sys, time, glob
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from speechless_amd import _lib
lib = _lib.lib()
pr = torch.cuda.get_device_properties(0)
bdf = "{:04x}:{:02x}:{:02x}.0".format(pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
hw = glob.glob("/sys/bus/pci/devices/{}/hwmon/hwmon*".format(bdf))[0]
def rd(n):
    try: return float(open(hw + "/" + n).read())
    except Exception: return float("nan")
b, t, k, lmax = 8, 4000, 29, 200
rng = np.random.RandomState(0)
dev = "cuda:0"
logits = torch.tensor(rng.randn(b, t, k).astype(np.float32), device=dev)
probs = torch.zeros((b, t, k), dtype=torch.float32, device=dev); logq = torch.zeros_like(probs)
lab_len = rng.randint(20, lmax + 1, size=b).astype(np.int32)
labels = np.zeros((b, lmax), dtype=np.int32)
for i, n in enumerate(lab_len): labels[i, :n] = rng.randint(0, k - 1, size=n)
lab = torch.tensor(labels, device=dev); ll = torch.tensor(lab_len, device=dev)
il = torch.full((b,), t, dtype=torch.int32, device=dev)
loss = torch.zeros((b,), dtype=torch.float32, device=dev)
dl = torch.zeros((b, t, 128), dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.call("sl_softmax_logq", logits.data_ptr(), probs.data_ptr(), logq.data_ptr(), b, t, k, k, t * k, 1e-8, st)
need = lib.raw("sl_ctc_workspace_bytes")(b, t, lmax)
ws = torch.empty((need,), dtype=torch.uint8, device=dev)
def run():
    lib.call("sl_ctc_loss_grad", probs.data_ptr(), logq.data_ptr(), lab.data_ptr(), ll.data_ptr(), il.data_ptr(),
             loss.data_ptr(), dl.data_ptr(), b, t, k, lmax, 0, 128, t * 128, _lib.SL_BF16, 1e-8, 1.0 / b, ws.data_ptr(), need, st)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
for name, fn in (("ctc only", run), ("ctc + a GEMM in between", lambda: (run(), torch.matmul(a, a)))):
    for sec in range(4):
        n = 0; t0 = time.perf_counter(); fs = []; ps = []
        while time.perf_counter() - t0 < 1.0:
            for _ in range(10): fn()
            torch.cuda.synchronize(); n += 10
            fs.append(rd("freq1_input") / 1e6); ps.append(rd("power1_input") / 1e6)
        print(name, "%.1f us/iter  sclk %.0f MHz  power %.0f W" % ((time.perf_counter() - t0) / n * 1e6, np.mean(fs), np.mean(ps)), flush=True)
